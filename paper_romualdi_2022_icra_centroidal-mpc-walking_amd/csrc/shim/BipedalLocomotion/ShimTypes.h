// Minimal stand-ins for the Eigen / BipedalLocomotionFramework types that appear in the signature of
// BipedalLocomotion::ReducedModelControllers::CentroidalMPC.  None of Eigen, BLF, manif, YARP is
// installed in the build image, so the facade (include/BipedalLocomotion/ReducedModelControllers/
// CentroidalMPC.h) can only be compile-checked against these.  Define CMPC_USE_REAL_BLF_HEADERS when
// the real headers are on the include path and this file is skipped.  Only the members the walking
// application touches on this path are modelled (file:line = where the reference uses them).
#pragma once
#ifndef CMPC_USE_REAL_BLF_HEADERS
#include <array>
#include <chrono>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace Eigen {
struct Vector3d {  // POD 3-vector with operator[] / operator()
    double v[3]{0, 0, 0};
    Vector3d() = default;
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};
template <class T> using Ref = T&;  // Eigen::Ref<const Vector3d> -> const Vector3d&
}  // namespace Eigen

namespace BipedalLocomotion {
namespace Math {
struct Wrenchd {  // force(3), torque(3); CentroidalMPCBlock.h:37
    Eigen::Vector3d f, t;
    const Eigen::Vector3d& force() const { return f; }
    const Eigen::Vector3d& torque() const { return t; }
};
}  // namespace Math

namespace ParametersHandler {
// subset of IParametersHandler used by initialize(): typed getParameter + getGroup
struct IParametersHandler {
    using shared_ptr = std::shared_ptr<IParametersHandler>;
    using weak_ptr = std::weak_ptr<const IParametersHandler>;
    virtual ~IParametersHandler() = default;
    virtual bool getParameter(const std::string& name, int& v) const = 0;
    virtual bool getParameter(const std::string& name, double& v) const = 0;
    virtual bool getParameter(const std::string& name, bool& v) const = 0;
    virtual bool getParameter(const std::string& name, std::string& v) const = 0;
    virtual bool getParameter(const std::string& name, std::vector<double>& v) const = 0;
    virtual weak_ptr getGroup(const std::string& name) const = 0;
};
}  // namespace ParametersHandler

namespace Contacts {
struct SE3d {  // manif::SE3d stand-in: rotation (row-major) + translation
    std::array<double, 9> R{1, 0, 0, 0, 1, 0, 0, 0, 1};
    Eigen::Vector3d p;
    const Eigen::Vector3d& translation() const { return p; }
    void translation(const Eigen::Vector3d& t) { p = t; }
};
struct PlannedContact {  // CentroidalMPCBlock.cpp:79-82 uses activationTime / deactivationTime / pose
    SE3d pose;
    std::chrono::nanoseconds activationTime{0}, deactivationTime{0};
    std::string name;
    int index{-1};
};
struct Corner { Eigen::Vector3d position, force; };  // WholeBodyQPBlock.cpp:824-829
struct DiscreteGeometryContact {                      // WholeBodyQPBlock.h:174-176
    SE3d pose;
    std::vector<Corner> corners;
    std::string name;
    int index{-1};
};
class ContactList {
    std::vector<PlannedContact> m_c;
public:
    using const_iterator = std::vector<PlannedContact>::const_iterator;
    bool addContact(const PlannedContact& c) { m_c.push_back(c); return true; }
    const_iterator cbegin() const { return m_c.cbegin(); }
    const_iterator cend() const { return m_c.cend(); }
    const_iterator begin() const { return m_c.cbegin(); }
    const_iterator end() const { return m_c.cend(); }
    std::size_t size() const { return m_c.size(); }
    PlannedContact& at(std::size_t i) { return m_c[i]; }
    const_iterator getActiveContact(const std::chrono::nanoseconds& t) const {
        for (auto it = m_c.cbegin(); it != m_c.cend(); ++it)
            if (it->activationTime <= t && t < it->deactivationTime) return it;
        return m_c.cend();
    }
    const_iterator getNextContact(const std::chrono::nanoseconds& t) const {
        for (auto it = m_c.cbegin(); it != m_c.cend(); ++it)
            if (it->activationTime > t) return it;
        return m_c.cend();
    }
};
using ContactListMap = std::map<std::string, ContactList>;
class ContactPhaseList {  // CentroidalMPCBlock.cpp:41,60,106 use lists() / setLists()
    ContactListMap m_lists;
public:
    const ContactListMap& lists() const { return m_lists; }
    bool setLists(const ContactListMap& l) { m_lists = l; return true; }
    ContactListMap& mutableLists() { return m_lists; }
};
}  // namespace Contacts
}  // namespace BipedalLocomotion
#endif  // CMPC_USE_REAL_BLF_HEADERS
