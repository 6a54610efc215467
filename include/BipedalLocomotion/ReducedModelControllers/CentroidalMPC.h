// Drop-in for <BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h> on the path the walking
// application uses (src/centroidal-mpc-walking/include/CentroidalMPCWalking/CentroidalMPCBlock.h:23,72;
// calls at src/CentroidalMPCBlock.cpp:144, 407, 579, 609, 615, 622): same namespace, class name, method
// names, argument meaning and bool-return / no-exception error convention, implemented as a batch = 1
// facade over the C ABI of libcmpc_hip.so (include/cmpc.h).  Header-only; link with -lcmpc_hip.
//
// With the real Eigen/BLF headers on the include path define CMPC_USE_REAL_BLF_HEADERS before including
// this file; otherwise the shim types of csrc/shim/BipedalLocomotion/ShimTypes.h are used (the build
// image has neither Eigen nor BLF, so "links unchanged" is asserted by signature, not by linking).
#pragma once

#ifdef CMPC_USE_REAL_BLF_HEADERS
#include <Eigen/Dense>
#include <BipedalLocomotion/Contacts/ContactPhaseList.h>
#include <BipedalLocomotion/Math/Wrench.h>
#include <BipedalLocomotion/ParametersHandler/IParametersHandler.h>
#else
#include <BipedalLocomotion/ShimTypes.h>
#endif

#include <chrono>
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "cmpc.h"

namespace BipedalLocomotion {
namespace ReducedModelControllers {

// what getOutput() returns (CentroidalMPCBlock.h:45; consumed at WholeBodyQPBlock.cpp:824-829,
// 1084, 1319-1335 and CentroidalMPCBlock.cpp:598, 626)
struct CentroidalMPCOutput {
    std::map<std::string, Contacts::DiscreteGeometryContact> contacts;  // first-knot corner forces + pose
    Contacts::ContactPhaseList contactPhaseList;                         // input list, next contact adjusted
    std::vector<Eigen::Vector3d> comTrajectory;                          // N+1 knots of the optimised CoM
};

class CentroidalMPC {
public:
    CentroidalMPC() = default;
    CentroidalMPC(const CentroidalMPC&) = delete;
    CentroidalMPC& operator=(const CentroidalMPC&) = delete;
    ~CentroidalMPC() { if (m_h) cmpc_destroy(m_h); }

    // keys: config/robots/<robot>/centroidal_mpc.ini (both generations of key names)
    bool initialize(std::weak_ptr<const ParametersHandler::IParametersHandler> handler)
    {
        auto ptr = handler.lock();
        if (!ptr) return err("[CentroidalMPC::initialize] invalid parameter handler");
        cmpc_default_config(&m_cfg);
        double dt = 0, horizon = 0;
        int steps = 0;
        if (ptr->getParameter("sampling_time", dt) && ptr->getParameter("time_horizon", horizon)) {
            steps = (int)std::lround(horizon / dt);
        } else if (ptr->getParameter("controller_sampling_time", dt) && ptr->getParameter("controller_horizon", steps)) {
        } else return err("[CentroidalMPC::initialize] sampling_time/time_horizon (or controller_*) missing");
        m_cfg.sampling_time = dt;
        m_cfg.horizon = steps;
        int nc = 2, slices = 1;
        ptr->getParameter("number_of_maximum_contacts", nc);
        ptr->getParameter("number_of_slices", slices);
        if (nc != 2 || slices != 1) return err("[CentroidalMPC::initialize] only 2 contacts / 1 friction slice are supported");
        ptr->getParameter("static_friction_coefficient", m_cfg.friction_coefficient);
        std::vector<double> v;
        if (!ptr->getParameter("com_weight", v) || v.size() != 3) return err("[CentroidalMPC::initialize] com_weight missing");
        for (int i = 0; i < 3; ++i) m_cfg.com_weight[i] = v[i];
        if (!ptr->getParameter("force_rate_of_change_weight", v) || v.size() != 3) return err("[CentroidalMPC::initialize] force_rate_of_change_weight missing");
        for (int i = 0; i < 3; ++i) m_cfg.force_rate_of_change_weight[i] = v[i];
        if (!ptr->getParameter("contact_position_weight", m_cfg.contact_position_weight)) return err("[CentroidalMPC::initialize] contact_position_weight missing");
        if (!ptr->getParameter("angular_momentum_weight", m_cfg.angular_momentum_weight)) return err("[CentroidalMPC::initialize] angular_momentum_weight missing");
        if (!ptr->getParameter("contact_force_symmetry_weight", m_cfg.contact_force_symmetry_weight)) m_cfg.contact_force_symmetry_weight = 0.0;
        double tol = 0;
        int maxit = 0;
        if (ptr->getParameter("ipopt_tolerance", tol) && tol > 0) m_cfg.tolerance = tol < 1e-6 ? tol : 1e-6;  // never looser than parity needs
        if (ptr->getParameter("ipopt_max_iteration", maxit) && maxit > 0) m_cfg.max_iterations = maxit;
        ptr->getParameter("is_warm_start_enabled", m_warm);
        // [CONTACT_i] groups, ordered like the reference's std::map (by contact name)
        std::map<std::string, int> order;
        struct C { std::string name; double corners[4][3]; double up[3], lo[3]; } cs[2];
        for (int i = 0; i < 2; ++i) {
            auto g = ptr->getGroup("CONTACT_" + std::to_string(i)).lock();
            if (!g) return err("[CentroidalMPC::initialize] group CONTACT_" + std::to_string(i) + " missing");
            int ncorn = 0;
            if (!g->getParameter("contact_name", cs[i].name) || !g->getParameter("number_of_corners", ncorn) || ncorn != 4)
                return err("[CentroidalMPC::initialize] contact_name / number_of_corners (must be 4)");
            for (int j = 0; j < 4; ++j) {
                if (!g->getParameter("corner_" + std::to_string(j), v) || v.size() != 3) return err("[CentroidalMPC::initialize] corner missing");
                for (int a = 0; a < 3; ++a) cs[i].corners[j][a] = v[a];
            }
            if (!g->getParameter("bounding_box_upper_limit", v) || v.size() != 3) return err("[CentroidalMPC::initialize] bounding_box_upper_limit missing");
            for (int a = 0; a < 3; ++a) cs[i].up[a] = v[a];
            if (!g->getParameter("bounding_box_lower_limit", v) || v.size() != 3) return err("[CentroidalMPC::initialize] bounding_box_lower_limit missing");
            for (int a = 0; a < 3; ++a) cs[i].lo[a] = v[a];
            order[cs[i].name] = i;
        }
        int slot = 0;
        for (auto& kv : order) {
            const C& c = cs[kv.second];
            m_names[slot] = c.name;
            for (int j = 0; j < 4; ++j)
                for (int a = 0; a < 3; ++a) m_cfg.corners[slot][j][a] = c.corners[j][a];
            for (int a = 0; a < 3; ++a) { m_up[slot][a] = (float)c.up[a]; m_lo[slot][a] = (float)c.lo[a]; }
            ++slot;
        }
        if (m_h) { cmpc_destroy(m_h); m_h = nullptr; }
        if (cmpc_create(&m_cfg, 1, 0, &m_h) != CMPC_OK) return err(std::string("[CentroidalMPC::initialize] ") + cmpc_last_error(nullptr));
        m_N = steps;
        m_dt = dt;
        return true;
    }

    // com, dcom, angular momentum and wrench are mass-normalised by the caller (CentroidalMPCBlock.cpp:403-410);
    // the wrench is held constant over the horizon
    bool setState(Eigen::Ref<const Eigen::Vector3d> com, Eigen::Ref<const Eigen::Vector3d> dcom,
                  Eigen::Ref<const Eigen::Vector3d> angularMomentum, const Math::Wrenchd& externalWrench)
    {
        if (!m_h) return err("[CentroidalMPC::setState] not initialised");
        float st[9];
        for (int i = 0; i < 3; ++i) { st[i] = (float)com[i]; st[3 + i] = (float)dcom[i]; st[6 + i] = (float)angularMomentum[i]; }
        std::vector<float> w(6 * (size_t)m_N);
        for (int k = 0; k < m_N; ++k)
            for (int i = 0; i < 3; ++i) { w[6 * k + i] = (float)externalWrench.force()[i]; w[6 * k + 3 + i] = (float)externalWrench.torque()[i]; }
        return ok(cmpc_set_state(m_h, st, w.data()), "[CentroidalMPC::setState]");
    }
    bool setState(Eigen::Ref<const Eigen::Vector3d> com, Eigen::Ref<const Eigen::Vector3d> dcom,
                  Eigen::Ref<const Eigen::Vector3d> angularMomentum)
    {
        return setState(com, dcom, angularMomentum, Math::Wrenchd());
    }

    // N+1 knots each (CentroidalMPCBlock.cpp:230-235, 579)
    bool setReferenceTrajectory(const std::vector<Eigen::Vector3d>& com, const std::vector<Eigen::Vector3d>& angularMomentum)
    {
        if (!m_h) return err("[CentroidalMPC::setReferenceTrajectory] not initialised");
        if ((int)com.size() != m_N + 1 || (int)angularMomentum.size() != m_N + 1)
            return err("[CentroidalMPC::setReferenceTrajectory] expected " + std::to_string(m_N + 1) + " knots");
        std::vector<float> c(3 * (size_t)(m_N + 1)), h(3 * (size_t)(m_N + 1));
        for (int k = 0; k <= m_N; ++k)
            for (int i = 0; i < 3; ++i) { c[3 * k + i] = (float)com[k][i]; h[3 * k + i] = (float)angularMomentum[k][i]; }
        return ok(cmpc_set_reference(m_h, c.data(), h.data()), "[CentroidalMPC::setReferenceTrajectory]");
    }

    // samples the schedule at the MPC knots (rule: contacts.py / DESIGN.md; t = 0 is "now", i.e. the
    // earliest time for which every foot has an active or upcoming contact is taken from m_now)
    bool setContactPhaseList(const Contacts::ContactPhaseList& list)
    {
        if (!m_h) return err("[CentroidalMPC::setContactPhaseList] not initialised");
        const int N = m_N;
        std::vector<float> R(2 * (size_t)N * 9), up(2 * (size_t)N * 3), lo(2 * (size_t)N * 3), en(2 * (size_t)N), nom(2 * (size_t)(N + 1) * 3), cur(6);
        m_landKnot[0] = m_landKnot[1] = -1;
        for (int c = 0; c < 2; ++c) {
            auto it = list.lists().find(m_names[c]);
            if (it == list.lists().end() || it->second.size() == 0) return err("[CentroidalMPC::setContactPhaseList] no contact list for " + m_names[c]);
            const auto& cl = it->second;
            auto owner = [&](std::chrono::nanoseconds t, bool& active) {
                auto a = cl.getActiveContact(t);
                active = a != cl.cend();
                if (active) return a;
                auto n = cl.getNextContact(t);
                if (n != cl.cend()) return n;
                return cl.cend() - 1;
            };
            for (int k = 0; k < N; ++k) {
                const auto t = m_now + std::chrono::nanoseconds((long long)std::llround(k * m_dt * 1e9));
                bool act = false;
                auto o = owner(t, act);
                en[(size_t)c * N + k] = act ? 1.f : 0.f;
                for (int i = 0; i < 9; ++i) R[((size_t)c * N + k) * 9 + i] = (float)o->pose.R[i];
                for (int i = 0; i < 3; ++i) {
                    up[((size_t)c * N + k) * 3 + i] = m_up[c][i];
                    lo[((size_t)c * N + k) * 3 + i] = m_lo[c][i];
                    nom[((size_t)c * (N + 1) + k + 1) * 3 + i] = (float)o->pose.translation()[i];
                }
                if (k == 0)
                    for (int i = 0; i < 3; ++i) { nom[((size_t)c * (N + 1)) * 3 + i] = (float)o->pose.translation()[i]; cur[3 * c + i] = (float)o->pose.translation()[i]; }
                if (!act && m_landKnot[c] < 0) {
                    bool nextAct = false;
                    if (k + 1 < N) owner(m_now + std::chrono::nanoseconds((long long)std::llround((k + 1) * m_dt * 1e9)), nextAct);
                    if (k + 1 == N || nextAct) m_landKnot[c] = k + 1;
                }
            }
        }
        m_list = list;
        return ok(cmpc_set_contacts(m_h, R.data(), up.data(), lo.data(), en.data(), nom.data(), cur.data()), "[CentroidalMPC::setContactPhaseList]");
    }

    // present time of the schedule (the reference's block passes absolute-time lists, CentroidalMPCBlock.cpp:594)
    void setCurrentTime(std::chrono::nanoseconds now) { m_now = now; }

    bool advance()
    {
        if (!m_h) return err("[CentroidalMPC::advance] not initialised");
        m_valid = false;
        if (cmpc_set_initial_guess(m_h, nullptr, m_warm && m_haveSolution ? 1 : 0) != CMPC_OK) return err(cmpc_last_error(m_h));
        if (cmpc_advance(m_h) != CMPC_OK) return err(std::string("[CentroidalMPC::advance] ") + cmpc_last_error(m_h));
        m_haveSolution = true;
        float f0[24], p0[6], pn[6];
        int kn[2];
        if (cmpc_get_output(m_h, f0, p0, pn, kn) != CMPC_OK) return err(cmpc_last_error(m_h));
        m_out.contacts.clear();
        m_out.contactPhaseList = m_list;
        for (int c = 0; c < 2; ++c) {
            auto lit = m_list.lists().find(m_names[c]);
            auto act = lit->second.getActiveContact(m_now);
            if (act != lit->second.cend()) {  // only active contacts are reported (WholeBodyQPBlock.cpp:824)
                Contacts::DiscreteGeometryContact d;
                d.name = m_names[c];
                d.pose = act->pose;
                d.pose.translation(Eigen::Vector3d(p0[3 * c], p0[3 * c + 1], p0[3 * c + 2]));
                d.corners.resize(4);
                for (int j = 0; j < 4; ++j) {
                    d.corners[j].position = Eigen::Vector3d(m_cfg.corners[c][j][0], m_cfg.corners[c][j][1], m_cfg.corners[c][j][2]);
                    d.corners[j].force = Eigen::Vector3d(f0[12 * c + 3 * j], f0[12 * c + 3 * j + 1], f0[12 * c + 3 * j + 2]);
                }
                m_out.contacts[m_names[c]] = d;
            }
            if (kn[c] >= 0) {  // step adjustment: the next contact takes the optimised landing position
                auto& cl = m_out.contactPhaseList.mutableLists()[m_names[c]];
                for (std::size_t i = 0; i < cl.size(); ++i)
                    if (cl.at(i).activationTime > m_now) { cl.at(i).pose.translation(Eigen::Vector3d(pn[3 * c], pn[3 * c + 1], pn[3 * c + 2])); break; }
            }
        }
        m_valid = true;
        return true;
    }

    const CentroidalMPCOutput& getOutput() const { return m_out; }
    bool isOutputValid() const { return m_valid; }
    const std::string& lastError() const { return m_err; }

private:
    bool err(const std::string& m) { m_err = m; std::fprintf(stderr, "%s\n", m.c_str()); return false; }
    bool ok(int rc, const char* where) { return rc == CMPC_OK ? true : err(std::string(where) + " " + cmpc_last_error(m_h)); }

    cmpc_handle m_h{nullptr};
    cmpc_config m_cfg{};
    int m_N{0};
    double m_dt{0};
    bool m_warm{true}, m_valid{false}, m_haveSolution{false};
    std::string m_names[2], m_err;
    float m_up[2][3]{}, m_lo[2][3]{};
    int m_landKnot[2]{-1, -1};
    std::chrono::nanoseconds m_now{0};
    Contacts::ContactPhaseList m_list;
    CentroidalMPCOutput m_out;
};

}  // namespace ReducedModelControllers
}  // namespace BipedalLocomotion
