// Compile check of the drop-in class, driving it exactly the way CentroidalMPCWalking::CentroidalMPCBlock does
// (src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp): initialize :144; setInput -> setState :396-411; advance:
// setReferenceTrajectory :579, updateContactPhaseList :594-607 (the block's own function, restated here in the shape of
// :32-110 with the same BLF calls), setContactPhaseList :609, advance :615, getOutput :622/:626, clock += dT :631.
// No call the reference does not make.  Built with -fsyntax-only by __graft_entry__.build(); examples/facade_demo.cpp
// is the same body linked against libcmpc_hip.so and run.
#include <BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h>

namespace blf = BipedalLocomotion;

// shape of CentroidalMPCBlock.cpp:32-110
static bool updateContactPhaseList(const std::chrono::nanoseconds& currentTime, const blf::Contacts::ContactPhaseList& mannPhaseList,
                                   const blf::Contacts::ContactPhaseList& mpcPhaseList, blf::Contacts::ContactPhaseList& contactPhaseList)
{
    blf::Contacts::ContactListMap contactListMap;
    for (const auto& [name, contactList] : mannPhaseList.lists()) {
        for (auto mannIt = contactList.getNextContact(currentTime); mannIt != contactList.cend(); ++mannIt)
            if (!contactListMap[name].addContact(*mannIt)) return false;
        const auto& mpcList = mpcPhaseList.lists().at(name);
        auto mpcPresentContact = mpcList.getActiveContact(currentTime);
        if (mpcPresentContact == mpcList.cend()) continue;
        auto mannPresentContact = contactList.getActiveContact(currentTime);
        if (mannPresentContact == contactList.cend()) return false;
        auto contact = *mpcPresentContact;
        contact.activationTime = (*mannPresentContact).activationTime;
        contact.deactivationTime = (*mannPresentContact).deactivationTime;
        if (!contactListMap[name].addContact(contact)) return false;
    }
    contactPhaseList.setLists(contactListMap);
    return true;
}

struct Block {
    blf::ReducedModelControllers::CentroidalMPC m_controller;  // default-constructible member (CentroidalMPCBlock.h:72)
    blf::ReducedModelControllers::CentroidalMPCOutput m_output;
    std::chrono::nanoseconds m_dT{60000000}, m_absoluteTime{0};
    bool m_isFirstRun{true};

    bool initialize(std::weak_ptr<const blf::ParametersHandler::IParametersHandler> handler) { return m_controller.initialize(handler); }

    bool setInput(const Eigen::Vector3d& com, const Eigen::Vector3d& dcom, const Eigen::Vector3d& h, const blf::Math::Wrenchd& w)
    {
        return m_controller.setState(com, dcom, h, w);  // :407
    }

    bool advance(const std::vector<Eigen::Vector3d>& comRef, const std::vector<Eigen::Vector3d>& hRef,
                 const blf::Contacts::ContactPhaseList& mannContactPhaseList)
    {
        if (!m_controller.setReferenceTrajectory(comRef, hRef)) return false;  // :579
        blf::Contacts::ContactPhaseList contactPhaseList;
        if (!m_isFirstRun) {
            if (!updateContactPhaseList(m_absoluteTime, mannContactPhaseList, m_controller.getOutput().contactPhaseList, contactPhaseList))
                return false;  // :594-604
        } else contactPhaseList = mannContactPhaseList;  // :606
        if (!m_controller.setContactPhaseList(contactPhaseList)) return false;  // :609
        if (!m_controller.advance()) return false;                              // :615
        m_output = m_controller.getOutput();  // copy-assigned and stored by value (:622)
        const auto& adjusted = m_controller.getOutput().contactPhaseList;  // :626
        for (const auto& [name, contact] : m_output.contacts)  // how WholeBodyQPBlock.cpp:824-829, :1319-1335 read it
            for (const auto& corner : contact.corners)
                (void)(corner.force[2] + corner.position[0] + contact.pose.translation()[0] + contact.pose.quat().coeffs()[3] + contact.pose.rotation()(0, 0));
        m_isFirstRun = false;
        m_absoluteTime += m_dT;  // :631
        return !adjusted.lists().empty() && m_controller.isOutputValid();
    }
};

#ifndef CMPC_FACADE_DEMO
int main() { return 0; }
#endif
