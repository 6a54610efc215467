// Stand-in for <manif/manif.h>: manif::SE3d as the reference uses it (see ../README.md):
//   construction from (translation, quaternion)      CentroidalMPCBlock.cpp:345
//   translation()                                    CentroidalMPCBlock.cpp:323, :362; WholeBodyQPBlock.cpp:1322
//   rotation()                                       CentroidalMPCBlock.cpp:363
//   quat()                                           CentroidalMPCBlock.cpp:345; WholeBodyQPBlock.cpp:1324
// manif poses are immutable in their parts: a pose with another translation is built anew (:345).
#pragma once
#include <Eigen/Dense>

namespace manif {
class SE3d {
    Eigen::Vector3d m_t;
    Eigen::Quaterniond m_q;
public:
    SE3d() = default;
    SE3d(const Eigen::Vector3d& translation, const Eigen::Quaterniond& quat) : m_t(translation), m_q(quat) {}
    static SE3d Identity() { return SE3d(); }
    Eigen::Vector3d translation() const { return m_t; }
    Eigen::Matrix3d rotation() const { return m_q.toRotationMatrix(); }
    Eigen::Quaterniond quat() const { return m_q; }
};
}  // namespace manif
