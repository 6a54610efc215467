// Stand-in for <BipedalLocomotion/Math/Wrench.h>: Math::Wrenchd with force() / torque()
// (CentroidalMPCBlock.h:37; filled at WholeBodyQPBlock.cpp:993-1021).  See ../../README.md.
#pragma once
#include <Eigen/Dense>

namespace BipedalLocomotion {
namespace Math {
class Wrenchd {
    Eigen::Vector3d m_f, m_t;
public:
    Wrenchd() = default;
    Wrenchd(const Eigen::Vector3d& force, const Eigen::Vector3d& torque) : m_f(force), m_t(torque) {}
    const Eigen::Vector3d& force() const { return m_f; }
    Eigen::Vector3d& force() { return m_f; }
    const Eigen::Vector3d& torque() const { return m_t; }
    Eigen::Vector3d& torque() { return m_t; }
    static Wrenchd Zero() { return Wrenchd(); }
};
}  // namespace Math
}  // namespace BipedalLocomotion
