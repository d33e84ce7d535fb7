// host_stub.cpp -- what csrc/jpeg.cpp needs from the rest of the library (the thread-local error string) when it is compiled alone into the study tools.
#include <string>
namespace sf { std::string& last_error_ref() { static thread_local std::string s; return s; } int usable_cpus() { return 8; } }
extern "C" const char* sf_last_error(void) { return sf::last_error_ref().c_str(); }
