// Compile check of the drop-in class against the shim headers, driving it the way
// CentroidalMPCWalking::CentroidalMPCBlock does (src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:
// initialize :144, setState :407, setReferenceTrajectory :579, setContactPhaseList :609, advance :615,
// getOutput :622/:626).  Built with -fsyntax-only by __graft_entry__.build(); examples/ links it.
#include <BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h>

namespace blf = BipedalLocomotion;

struct Block {
    blf::ReducedModelControllers::CentroidalMPC m_controller;  // default-constructible member (CentroidalMPCBlock.h:72)
    blf::ReducedModelControllers::CentroidalMPCOutput m_output;

    bool tick(std::weak_ptr<const blf::ParametersHandler::IParametersHandler> handler, const blf::Contacts::ContactPhaseList& list)
    {
        if (!m_controller.initialize(handler)) return false;
        Eigen::Vector3d com(0, 0, 0.7), dcom, h;
        blf::Math::Wrenchd w;
        if (!m_controller.setState(com, dcom, h, w)) return false;
        std::vector<Eigen::Vector3d> comRef(21, com), hRef(21);
        if (!m_controller.setReferenceTrajectory(comRef, hRef)) return false;
        if (!m_controller.setContactPhaseList(list)) return false;
        if (!m_controller.advance()) return false;
        m_output = m_controller.getOutput();  // copy-assigned and stored by value (CentroidalMPCBlock.cpp:622)
        const auto& contactPhaseList = m_controller.getOutput().contactPhaseList;  // :598, :626
        for (const auto& [name, contact] : m_output.contacts)
            for (const auto& corner : contact.corners) (void)(corner.force[2] + corner.position[0] + contact.pose.translation()[0]);
        return !contactPhaseList.lists().empty() && m_controller.isOutputValid();
    }
};

int main() { return 0; }
