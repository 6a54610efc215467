// Links the drop-in class (include/BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h) against libcmpc_hip.so
// and runs a short walk the way the reference's block drives it (the Block of csrc/facade_check.cpp: setInput ->
// setState, then setReferenceTrajectory / updateContactPhaseList / setContactPhaseList / advance / getOutput, clock
// += dT), 12 ticks across a lift-off, with the class's own static merge cross-checked against the block's function.
// Build (tests/test_gpu_facade.py does this):
//   g++ -std=c++17 -I include -I <pkg>/csrc/shim examples/facade_demo.cpp -L <pkg> -lcmpc_hip -Wl,-rpath,<pkg>
#define CMPC_FACADE_DEMO
#include "../paper_romualdi_2022_icra_centroidal-mpc-walking_amd/csrc/facade_check.cpp"

#include <cstdio>
#include <map>

using blf::ParametersHandler::IParametersHandler;

struct MapHandler : IParametersHandler {
    std::map<std::string, double> d;
    std::map<std::string, std::string> s;
    std::map<std::string, std::vector<double>> v;
    std::map<std::string, std::shared_ptr<MapHandler>> g;
    bool getParameter(const std::string& n, int& o) const override { auto i = d.find(n); if (i == d.end()) return false; o = (int)i->second; return true; }
    bool getParameter(const std::string& n, double& o) const override { auto i = d.find(n); if (i == d.end()) return false; o = i->second; return true; }
    bool getParameter(const std::string& n, bool& o) const override { auto i = d.find(n); if (i == d.end()) return false; o = i->second != 0; return true; }
    bool getParameter(const std::string& n, std::string& o) const override { auto i = s.find(n); if (i == s.end()) return false; o = i->second; return true; }
    bool getParameter(const std::string& n, std::vector<double>& o) const override { auto i = v.find(n); if (i == v.end()) return false; o = i->second; return true; }
    weak_ptr getGroup(const std::string& n) const override { auto i = g.find(n); return i == g.end() ? weak_ptr() : weak_ptr(i->second); }
};

int main()
{
    // config/robots/ergoCubGazeboV1/centroidal_mpc.ini:3-42
    auto h = std::make_shared<MapHandler>();
    h->d = {{"sampling_time", 0.06}, {"time_horizon", 1.2}, {"number_of_maximum_contacts", 2}, {"number_of_slices", 1},
            {"static_friction_coefficient", 0.33}, {"is_warm_start_enabled", 1}, {"ipopt_tolerance", 1e-4},
            {"contact_position_weight", 2e3}, {"angular_momentum_weight", 1e2}, {"contact_force_symmetry_weight", 100.0}};
    h->v = {{"com_weight", {10, 10, 200}}, {"force_rate_of_change_weight", {10, 10, 10}}};
    const char* names[2] = {"left_foot", "right_foot"};
    for (int i = 0; i < 2; ++i) {
        auto c = std::make_shared<MapHandler>();
        c->d = {{"number_of_corners", 4}};
        c->s = {{"contact_name", names[i]}};
        c->v = {{"corner_0", {0.08, 0.01, 0}}, {"corner_1", {0.08, -0.01, 0}}, {"corner_2", {-0.08, -0.01, 0}}, {"corner_3", {-0.08, 0.01, 0}},
                {"bounding_box_upper_limit", i == 0 ? std::vector<double>{0.01, 0.05, 0} : std::vector<double>{0.01, 0.0, 0}},
                {"bounding_box_lower_limit", i == 0 ? std::vector<double>{-0.01, 0.0, 0} : std::vector<double>{-0.01, -0.05, 0}}};
        h->g["CONTACT_" + std::to_string(i)] = c;
    }
    Block block;
    if (!block.initialize(h)) return 1;

    // the planner's list: the left foot lifts at 0.36 s and lands 0.1 m ahead at 0.84 s; absolute times from zero
    using namespace std::chrono_literals;
    blf::Contacts::ContactListMap lists;
    blf::Contacts::PlannedContact c;
    c.name = "left_foot"; c.pose = manif::SE3d(Eigen::Vector3d(0, 0.08, 0), Eigen::Quaterniond(1, 0, 0, 0)); c.activationTime = 0s; c.deactivationTime = 360ms;
    lists["left_foot"].addContact(c);
    c.pose = manif::SE3d(Eigen::Vector3d(0.1, 0.08, 0), Eigen::Quaterniond(1, 0, 0, 0)); c.activationTime = 840ms; c.deactivationTime = 100s;
    lists["left_foot"].addContact(c);
    c.name = "right_foot"; c.pose = manif::SE3d(Eigen::Vector3d(0, -0.08, 0), Eigen::Quaterniond(1, 0, 0, 0)); c.activationTime = 0s; c.deactivationTime = 100s;
    lists["right_foot"].addContact(c);
    blf::Contacts::ContactPhaseList mann;
    mann.setLists(lists);

    Eigen::Vector3d com(0.01, -0.005, 0.69), dcom(0.02, 0, 0), ang;
    blf::Math::Wrenchd w;
    std::vector<Eigen::Vector3d> comRef(21, Eigen::Vector3d(0.03, -0.02, 0.7)), hRef(21);
    double fz = 0;
    for (int tick = 0; tick < 12; ++tick) {
        if (tick > 0) {  // the class's own merge must rebuild what the block's function builds
            blf::Contacts::ContactPhaseList a, b;
            const bool ra = updateContactPhaseList(block.m_absoluteTime, mann, block.m_controller.getOutput().contactPhaseList, a);
            const bool rb = blf::ReducedModelControllers::CentroidalMPC::mergeContactPhaseLists(block.m_absoluteTime, mann, block.m_controller.getOutput().contactPhaseList, b);
            if (ra != rb) { std::fprintf(stderr, "merge mismatch (return) at tick %d\n", tick); return 3; }
            for (const auto& [name, la] : a.lists()) {
                const auto& lb = b.lists().at(name);
                if (la.size() != lb.size()) { std::fprintf(stderr, "merge mismatch (size) at tick %d\n", tick); return 3; }
                auto ib = lb.cbegin();
                for (auto ia = la.cbegin(); ia != la.cend(); ++ia, ++ib)
                    if (ia->activationTime != ib->activationTime || ia->deactivationTime != ib->deactivationTime
                        || std::fabs(ia->pose.translation()[0] - ib->pose.translation()[0]) > 1e-6
                        || std::fabs(ia->pose.translation()[1] - ib->pose.translation()[1]) > 1e-6) { std::fprintf(stderr, "merge mismatch at tick %d\n", tick); return 3; }
            }
        }
        if (!block.setInput(com, dcom, ang, w) || !block.advance(comRef, hRef, mann)) {
            std::fprintf(stderr, "tick %d failed: %s\n", tick, block.m_controller.lastError().c_str());
            return 2;
        }
        // feed the first predicted state back as the next measurement (a stand-in for the whole-body loop)
        const auto& out = block.m_output;
        com = out.comTrajectory[1];
        fz = 0;
        int ncontacts = 0;
        for (const auto& [name, contact] : out.contacts) {
            ++ncontacts;
            for (const auto& corner : contact.corners) fz += corner.force[2];
        }
        std::printf("tick %d contacts %d total_fz %.6f com %.5f %.5f %.5f\n", tick, ncontacts, fz, com[0], com[1], com[2]);
        // the numbers themselves (tests/test_gpu_facade.py replays the ticks through the C ABI and compares): per contact in the
        // output map the 12 first-knot corner-force components (what WholeBodyQPBlock.cpp:824-829 reads), then the CoM trajectory
        for (const auto& [name, contact] : out.contacts) {
            std::printf("forces %d %s", tick, name.c_str());
            for (const auto& corner : contact.corners) std::printf(" %.9g %.9g %.9g", corner.force[0], corner.force[1], corner.force[2]);
            std::printf("\n");
        }
        std::printf("comtraj %d", tick);
        for (const auto& ck : out.comTrajectory) std::printf(" %.9g %.9g %.9g", ck[0], ck[1], ck[2]);
        std::printf("\n");
    }
    const auto& ll = block.m_controller.getOutput().contactPhaseList.lists().at("left_foot");
    const auto next = ll.getNextContact(block.m_absoluteTime - block.m_dT);
    if (next == ll.cend()) { std::fprintf(stderr, "no next contact in the adjusted list\n"); return 4; }
    std::printf("total_fz %.6f\nnext_left %.6f %.6f %.6f\nticks 12\n", fz, next->pose.translation()[0], next->pose.translation()[1], next->pose.translation()[2]);
    return 0;
}
