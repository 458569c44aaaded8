// Command line of the reference's modules as the workflow scripts pass it: positional DB paths followed by
// "--flag value" pairs (M/src/commons/Parameters.cpp parseParameters).  Boolean flags toggle when no value follows
// (Parameters.cpp:1920-1928); MultiParam values come as "aa:11,nucl:5" or "seq:..,prof:.." or a bare value.
#ifndef SD_ARGS_H
#define SD_ARGS_H

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace sdcli {

struct Args {
    std::string module;
    std::vector<std::string> pos;
    std::map<std::string, std::string> opt;

    // returns false + message on an unrecognised flag or a missing value
    bool parse(int argc, const char **argv, std::string *err);

    bool has(const std::string &f) const { return opt.count(f) != 0; }
    std::string str(const std::string &f, const std::string &def) const;
    long long integer(const std::string &f, long long def) const;
    double real(const std::string &f, double def) const;
    bool flag(const std::string &f, bool def) const;
    // "aa:11,nucl:5" -> 11 ; "seq:2147483647,prof:..." -> seq value; a bare value is returned as is
    std::string multi(const std::string &f, const std::string &tag, const std::string &def) const;
    // the tail of argv in "--flag value" form, e.g. to forward it to a sub-step
    std::vector<std::string> flagsAsArgv() const;
};

}  // namespace sdcli
#endif
