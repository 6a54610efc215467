// Links the drop-in class (include/BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h) against
// libcmpc_hip.so and runs one MPC tick the way CentroidalMPCBlock does.  Prints the first-knot
// vertical force per foot and the adjusted next footstep.  Build (tests/test_gpu_facade.py does this):
//   g++ -std=c++17 -I include -I <pkg>/csrc/shim examples/facade_demo.cpp -L <pkg> -lcmpc_hip -Wl,-rpath,<pkg>
#include <BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h>

#include <cstdio>
#include <map>

namespace blf = BipedalLocomotion;
using blf::ParametersHandler::IParametersHandler;

struct MapHandler : IParametersHandler, std::enable_shared_from_this<MapHandler> {
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
    blf::ReducedModelControllers::CentroidalMPC mpc;
    if (!mpc.initialize(h)) return 1;

    using namespace std::chrono_literals;
    blf::Contacts::ContactListMap lists;
    blf::Contacts::PlannedContact c;
    c.name = "left_foot"; c.pose.p = Eigen::Vector3d(0, 0.08, 0); c.activationTime = -1s; c.deactivationTime = 360ms;
    lists["left_foot"].addContact(c);
    c.pose.p = Eigen::Vector3d(0.1, 0.08, 0); c.activationTime = 840ms; c.deactivationTime = 100s;
    lists["left_foot"].addContact(c);
    c.name = "right_foot"; c.pose.p = Eigen::Vector3d(0, -0.08, 0); c.activationTime = -1s; c.deactivationTime = 100s;
    lists["right_foot"].addContact(c);
    blf::Contacts::ContactPhaseList list;
    list.setLists(lists);

    Eigen::Vector3d com(0.01, -0.005, 0.69), dcom(0.02, 0, 0), ang;
    blf::Math::Wrenchd w;
    std::vector<Eigen::Vector3d> comRef(21, Eigen::Vector3d(0, 0, 0.7)), hRef(21);
    for (int tick = 0; tick < 2; ++tick) {  // second tick exercises the warm start
        if (!mpc.setState(com, dcom, ang, w) || !mpc.setReferenceTrajectory(comRef, hRef) || !mpc.setContactPhaseList(list) || !mpc.advance()) {
            std::fprintf(stderr, "tick failed: %s\n", mpc.lastError().c_str());
            return 2;
        }
    }
    const auto& out = mpc.getOutput();
    double fz = 0;
    for (const auto& [name, contact] : out.contacts) {
        double f = 0;
        for (const auto& corner : contact.corners) f += corner.force[2];
        std::printf("contact %s fz %.6f\n", name.c_str(), f);
        fz += f;
    }
    const auto& next = *(out.contactPhaseList.lists().at("left_foot").cbegin() + 1);
    std::printf("total_fz %.6f\nnext_left %.6f %.6f %.6f\n", fz, next.pose.translation()[0], next.pose.translation()[1], next.pose.translation()[2]);
    return 0;
}
