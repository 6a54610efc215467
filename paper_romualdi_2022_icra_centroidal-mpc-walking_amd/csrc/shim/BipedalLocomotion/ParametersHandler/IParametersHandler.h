// Stand-in for <BipedalLocomotion/ParametersHandler/IParametersHandler.h>: typed getParameter + getGroup, the
// part of the interface initialize() uses (the reference reads its own keys the same way, CentroidalMPCBlock.cpp:118-160).
// See ../../README.md.
#pragma once
#include <memory>
#include <string>
#include <vector>

namespace BipedalLocomotion {
namespace ParametersHandler {
class IParametersHandler {
public:
    using shared_ptr = std::shared_ptr<IParametersHandler>;
    using weak_ptr = std::weak_ptr<IParametersHandler>;
    virtual ~IParametersHandler() = default;
    virtual bool getParameter(const std::string& name, int& v) const = 0;
    virtual bool getParameter(const std::string& name, double& v) const = 0;
    virtual bool getParameter(const std::string& name, bool& v) const = 0;
    virtual bool getParameter(const std::string& name, std::string& v) const = 0;
    virtual bool getParameter(const std::string& name, std::vector<double>& v) const = 0;
    virtual weak_ptr getGroup(const std::string& name) const = 0;
};
}  // namespace ParametersHandler
}  // namespace BipedalLocomotion
