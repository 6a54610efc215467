// Stand-in for <BipedalLocomotion/Contacts/ContactPhaseList.h> (and Contact.h / ContactList.h it pulls in): the
// members the reference touches on the CentroidalMPC path.  See ../../README.md.
//   PlannedContact: pose, activationTime, deactivationTime, name, index        CentroidalMPCBlock.cpp:79-81, :350-360
//   ContactList: addContact, getNextContact, getActiveContact, cbegin / cend   CentroidalMPCBlock.cpp:44-48, :61, :69
//                (an ordered set: its iterators are bidirectional, not random access)
//   ContactPhaseList: lists(), setLists()                                      CentroidalMPCBlock.cpp:41, :60, :107
//   DiscreteGeometryContact: pose, corners[].position / .force, name, index    WholeBodyQPBlock.cpp:824-829, :1319-1335
#pragma once
#include <chrono>
#include <map>
#include <set>
#include <string>
#include <vector>

#include <manif/manif.h>

namespace BipedalLocomotion {
namespace Contacts {

struct ContactBase {
    manif::SE3d pose;
    int index{-1};
    std::string name;
};
struct PlannedContact : public ContactBase {
    std::chrono::nanoseconds activationTime{std::chrono::nanoseconds::zero()};
    std::chrono::nanoseconds deactivationTime{std::chrono::nanoseconds::max()};
};
struct Corner {
    Eigen::Vector3d position;
    Eigen::Vector3d force;
};
struct DiscreteGeometryContact : public ContactBase {
    std::vector<Corner> corners;
};

class ContactList {
    struct Before {
        bool operator()(const PlannedContact& a, const PlannedContact& b) const { return a.deactivationTime <= b.activationTime; }
    };
    std::set<PlannedContact, Before> m_contacts;  // ordered by time, contacts of one foot do not overlap
public:
    using const_iterator = std::set<PlannedContact, Before>::const_iterator;
    bool addContact(const PlannedContact& c)
    {
        if (c.deactivationTime < c.activationTime) return false;
        return m_contacts.insert(c).second;  // an overlapping contact compares equivalent: refused, as in BLF
    }
    const_iterator cbegin() const { return m_contacts.cbegin(); }
    const_iterator cend() const { return m_contacts.cend(); }
    const_iterator begin() const { return m_contacts.cbegin(); }
    const_iterator end() const { return m_contacts.cend(); }
    std::size_t size() const { return m_contacts.size(); }
    // the contact with activationTime <= t < deactivationTime, or cend()
    const_iterator getActiveContact(const std::chrono::nanoseconds& t) const
    {
        for (auto it = m_contacts.cbegin(); it != m_contacts.cend(); ++it)
            if (it->activationTime <= t && t < it->deactivationTime) return it;
        return m_contacts.cend();
    }
    // the contact with the lowest activationTime strictly after t, or cend()
    const_iterator getNextContact(const std::chrono::nanoseconds& t) const
    {
        for (auto it = m_contacts.cbegin(); it != m_contacts.cend(); ++it)
            if (it->activationTime > t) return it;
        return m_contacts.cend();
    }
};
using ContactListMap = std::map<std::string, ContactList>;

class ContactPhaseList {
    ContactListMap m_lists;
public:
    const ContactListMap& lists() const { return m_lists; }
    bool setLists(const ContactListMap& l) { m_lists = l; return true; }
};

}  // namespace Contacts
}  // namespace BipedalLocomotion
