// Stand-in for <yaml-cpp/yaml.h>: utils/options.h only names YAML::Node in declarations; utils/options.cc (the one
// file that reads YAML) is not compiled by oracle/Makefile.ref.  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_STUBS_YAML_H_
#define ORACLE_REF_STUBS_YAML_H_
namespace YAML {
class Node;
}
#endif
