// Drop-in for <BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h> on the path the walking
// application uses (src/centroidal-mpc-walking/include/CentroidalMPCWalking/CentroidalMPCBlock.h:23,72;
// calls at src/CentroidalMPCBlock.cpp:144, 407, 579, 609, 615, 622): same namespace, class name, method
// names, argument meaning and bool-return / no-exception error convention, implemented as a batch = 1
// facade over the C ABI of libcmpc_hip.so (include/cmpc.h).  Header-only; link with -lcmpc_hip.
//
// It uses only the Eigen / manif / BLF calls the reference itself makes (pose.translation(), pose.rotation(),
// pose.quat(), manif::SE3d(translation, quat), ContactList::addContact / getActiveContact / getNextContact /
// cbegin / cend, ContactPhaseList::lists / setLists; CentroidalMPCBlock.cpp:41-107, :345, :362-363).  The build
// image has none of those libraries: csrc/shim/ holds stand-in headers at the same include paths that declare
// exactly those members, so this header is compiled and run against them unchanged (no switch, no #ifdef).
//
// Time: the reference never tells the controller the time; its block passes absolute-time contact lists and
// advances its own clock by the sampling time per tick (CentroidalMPCBlock.cpp:631).  The class does the same:
// its clock starts at zero and advances by dt after every successful advance().
#pragma once

#include <Eigen/Dense>
#include <manif/manif.h>

#include <BipedalLocomotion/Contacts/ContactPhaseList.h>
#include <BipedalLocomotion/Math/Wrench.h>
#include <BipedalLocomotion/ParametersHandler/IParametersHandler.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <iterator>
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

    // com, dcom, angular momentum and wrench are mass-normalised by the caller (CentroidalMPCBlock.cpp:403-410).
    // The measured wrench enters the first knot only, the rest of the horizon sees none (how BLF spreads it is not
    // visible from the reference tree -- parity unpinned; a persistent push is expressed through the C ABI, which
    // takes the wrench per knot)
    bool setState(Eigen::Ref<const Eigen::Vector3d> com, Eigen::Ref<const Eigen::Vector3d> dcom,
                  Eigen::Ref<const Eigen::Vector3d> angularMomentum, const Math::Wrenchd& externalWrench)
    {
        if (!m_h) return err("[CentroidalMPC::setState] not initialised");
        float st[9];
        for (int i = 0; i < 3; ++i) { st[i] = (float)com[i]; st[3 + i] = (float)dcom[i]; st[6 + i] = (float)angularMomentum[i]; }
        std::vector<float> w(6 * (size_t)m_N, 0.f);
        for (int i = 0; i < 3; ++i) { w[i] = (float)externalWrench.force()[i]; w[3 + i] = (float)externalWrench.torque()[i]; }
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

    // samples the schedule at the MPC knots now + k dt (rule: cmpc_contacts_sample, include/cmpc.h)
    bool setContactPhaseList(const Contacts::ContactPhaseList& list)
    {
        if (!m_h) return err("[CentroidalMPC::setContactPhaseList] not initialised");
        ListArrays a;
        if (!toArrays(list, a)) return false;
        m_list = list;
        m_arrays = a;
        const float up[6] = {m_up[0][0], m_up[0][1], m_up[0][2], m_up[1][0], m_up[1][1], m_up[1][2]};
        const float lo[6] = {m_lo[0][0], m_lo[0][1], m_lo[0][2], m_lo[1][0], m_lo[1][1], m_lo[1][2]};
        return ok(cmpc_set_contact_lists(m_h, a.M, seconds(m_now), a.t.data(), a.pose.data(), a.n, up, lo, m_landKnot),
                  "[CentroidalMPC::setContactPhaseList]");
    }

    bool advance()
    {
        if (!m_h) return err("[CentroidalMPC::advance] not initialised");
        m_valid = false;
        if (m_arrays.M == 0) return err("[CentroidalMPC::advance] setContactPhaseList has not been called");
        if (cmpc_set_initial_guess(m_h, nullptr, m_warm && m_haveSolution ? 1 : 0) != CMPC_OK) return err(cmpc_last_error(m_h));
        if (cmpc_advance(m_h) != CMPC_OK) return err(std::string("[CentroidalMPC::advance] ") + cmpc_last_error(m_h));
        m_haveSolution = true;
        float f0[24], p0[6], pn[6];
        int kn[2];
        if (cmpc_get_output(m_h, f0, p0, pn, kn) != CMPC_OK) return err(cmpc_last_error(m_h));
        std::vector<float> x((size_t)45 * m_N + 15);
        if (cmpc_get_solution(m_h, x.data(), nullptr) != CMPC_OK) return err(cmpc_last_error(m_h));
        m_out.contacts.clear();
        m_out.comTrajectory.resize((size_t)m_N + 1);
        for (int k = 0; k <= m_N; ++k) m_out.comTrajectory[k] = Eigen::Vector3d(x[3 * k], x[3 * k + 1], x[3 * k + 2]);
        // step adjustment on the arrays (cmpc_contacts_adjust), then the list is rebuilt contact by contact
        ListArrays adj = m_arrays;
        if (cmpc_contacts_adjust(m_N, 1, adj.M, seconds(m_now), x.data(), m_landKnot, adj.t.data(), adj.pose.data(), adj.n) != CMPC_OK)
            return err(cmpc_last_error(nullptr));
        Contacts::ContactListMap map;
        for (int c = 0; c < 2; ++c) {
            const Contacts::ContactList& cl = m_list.lists().at(m_names[c]);
            auto act = cl.getActiveContact(m_now);
            if (act != cl.cend()) {  // only active contacts are reported (WholeBodyQPBlock.cpp:824)
                Contacts::DiscreteGeometryContact d;
                d.name = m_names[c];
                d.index = act->index;
                d.pose = manif::SE3d(Eigen::Vector3d(p0[3 * c], p0[3 * c + 1], p0[3 * c + 2]), act->pose.quat());
                d.corners.resize(4);
                for (int j = 0; j < 4; ++j) {
                    d.corners[j].position = Eigen::Vector3d(m_cfg.corners[c][j][0], m_cfg.corners[c][j][1], m_cfg.corners[c][j][2]);
                    d.corners[j].force = Eigen::Vector3d(f0[12 * c + 3 * j], f0[12 * c + 3 * j + 1], f0[12 * c + 3 * j + 2]);
                }
                m_out.contacts[m_names[c]] = d;
            }
            Contacts::ContactList out;
            int m = 0;
            for (auto it = cl.cbegin(); it != cl.cend(); ++it, ++m) {
                Contacts::PlannedContact pc = *it;
                const float* q = adj.pose.data() + 7 * ((size_t)c * adj.M + m);
                pc.pose = manif::SE3d(Eigen::Vector3d(q[0], q[1], q[2]), it->pose.quat());
                if (!out.addContact(pc)) return err("[CentroidalMPC::advance] unable to rebuild the contact list of " + m_names[c]);
            }
            map[m_names[c]] = out;
        }
        if (!m_out.contactPhaseList.setLists(map)) return err("[CentroidalMPC::advance] unable to set the adjusted contact lists");
        m_now += std::chrono::nanoseconds((long long)std::llround(m_dt * 1e9));  // the caller's clock does the same (:631)
        m_valid = true;
        return true;
    }

    const CentroidalMPCOutput& getOutput() const { return m_out; }
    bool isOutputValid() const { return m_valid; }
    const std::string& lastError() const { return m_err; }

    // updateContactPhaseList of the reference's block (CentroidalMPCBlock.cpp:32-110) over the C ABI
    // (cmpc_contacts_merge): future contacts from the planner; the present contact keeps the pose the MPC gave it and
    // takes the planner's timing.  false when the planner has no active contact under an active MPC contact (:69-77).
    static bool mergeContactPhaseLists(const std::chrono::nanoseconds& currentTime, const Contacts::ContactPhaseList& plannerPhaseList,
                                       const Contacts::ContactPhaseList& mpcPhaseList, Contacts::ContactPhaseList& contactPhaseList)
    {
        std::vector<std::string> names;
        for (const auto& kv : plannerPhaseList.lists()) names.push_back(kv.first);
        if (names.size() != 2) return false;
        ListArrays p, q, o;
        if (!toArrays(plannerPhaseList, names, p) || !toArrays(mpcPhaseList, names, q)) return false;
        const int M = p.M > q.M ? p.M : q.M;
        p.resize(M); q.resize(M); o.resize(M);
        int good = 0;
        if (cmpc_contacts_merge(1, M, seconds(currentTime), p.t.data(), p.pose.data(), p.n, q.t.data(), q.pose.data(), q.n, o.t.data(),
                                o.pose.data(), o.n, &good) != CMPC_OK || !good)
            return false;
        Contacts::ContactListMap map;
        for (int c = 0; c < 2; ++c) {
            Contacts::ContactList out;
            for (int m = 0; m < o.n[c]; ++m) {
                const size_t e = (size_t)c * M + m;
                Contacts::PlannedContact pc;
                pc.name = names[c];
                pc.activationTime = std::chrono::nanoseconds((long long)std::llround(o.t[2 * e] * 1e9));
                pc.deactivationTime = o.t[2 * e + 1] >= 9e9 ? std::chrono::nanoseconds::max() : std::chrono::nanoseconds((long long)std::llround(o.t[2 * e + 1] * 1e9));
                const float* s = o.pose.data() + 7 * e;
                pc.pose = manif::SE3d(Eigen::Vector3d(s[0], s[1], s[2]), Eigen::Quaterniond(s[3], s[4], s[5], s[6]));
                if (!out.addContact(pc)) return false;
            }
            map[names[c]] = out;
        }
        return contactPhaseList.setLists(map);
    }

private:
    // a phase list as the arrays of the C ABI (include/cmpc.h, "contact schedules, batched"), batch = 1
    struct ListArrays {
        int M{0};
        int n[2]{0, 0};
        std::vector<double> t;
        std::vector<float> pose;
        void resize(int newM)
        {
            std::vector<double> nt((size_t)4 * newM, 0.0);
            std::vector<float> np((size_t)14 * newM, 0.f);
            for (int c = 0; c < 2; ++c)
                for (int m = 0; m < n[c] && m < newM; ++m) {
                    nt[2 * ((size_t)c * newM + m)] = t[2 * ((size_t)c * M + m)];
                    nt[2 * ((size_t)c * newM + m) + 1] = t[2 * ((size_t)c * M + m) + 1];
                    for (int i = 0; i < 7; ++i) np[7 * ((size_t)c * newM + m) + i] = pose[7 * ((size_t)c * M + m) + i];
                }
            t.swap(nt); pose.swap(np); M = newM;
        }
    };
    static double seconds(const std::chrono::nanoseconds& t)
    {
        return t == std::chrono::nanoseconds::max() ? 1e10 : (double)t.count() * 1e-9;
    }
    static bool toArrays(const Contacts::ContactPhaseList& list, const std::vector<std::string>& names, ListArrays& a)
    {
        std::size_t M = 1;
        for (const auto& nm : names) {
            auto it = list.lists().find(nm);
            if (it == list.lists().end()) return false;
            if (it->second.size() > M) M = it->second.size();
        }
        a = ListArrays();
        a.resize((int)M);
        for (int c = 0; c < 2; ++c) {
            const Contacts::ContactList& cl = list.lists().at(names[c]);
            int m = 0;
            for (auto it = cl.cbegin(); it != cl.cend(); ++it, ++m) {
                const size_t e = (size_t)c * M + m;
                a.t[2 * e] = seconds(it->activationTime);
                a.t[2 * e + 1] = seconds(it->deactivationTime);
                const Eigen::Vector3d p = it->pose.translation();
                const Eigen::Quaterniond q = it->pose.quat();
                float* s = a.pose.data() + 7 * e;
                s[0] = (float)p[0]; s[1] = (float)p[1]; s[2] = (float)p[2];
                s[3] = (float)q.w(); s[4] = (float)q.x(); s[5] = (float)q.y(); s[6] = (float)q.z();
            }
            a.n[c] = m;
        }
        return true;
    }
    bool toArrays(const Contacts::ContactPhaseList& list, ListArrays& a)
    {
        const std::vector<std::string> names{m_names[0], m_names[1]};
        for (int c = 0; c < 2; ++c) {
            auto it = list.lists().find(m_names[c]);
            if (it == list.lists().end() || it->second.size() == 0) return err("[CentroidalMPC::setContactPhaseList] no contact list for " + m_names[c]);
        }
        return toArrays(list, names, a) ? true : err("[CentroidalMPC::setContactPhaseList] malformed contact phase list");
    }

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
    ListArrays m_arrays;
    CentroidalMPCOutput m_out;
};

}  // namespace ReducedModelControllers
}  // namespace BipedalLocomotion
