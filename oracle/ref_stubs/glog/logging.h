// Stand-in for <glog/logging.h> (not installed here, no network): just enough for the reference's runtime/ sources to
// compile in place under oracle/Makefile.ref.  TEST INFRASTRUCTURE.  LOG(FATAL) and a failed CHECK throw, so that a test
// sees them instead of the process dying.
#ifndef ORACLE_REF_STUBS_GLOG_LOGGING_H_
#define ORACLE_REF_STUBS_GLOG_LOGGING_H_
#include <sstream>
#include <stdexcept>
namespace oracle_ref_stub {
struct LogLine {
  bool fatal;
  std::ostringstream text;
  explicit LogLine(bool f) : fatal(f) {}
  ~LogLine() noexcept(false) {
    if (fatal) throw std::runtime_error(text.str());
  }
  template <typename T>
  LogLine &operator<<(const T &v) {
    text << v;
    return *this;
  }
};
struct Voidify {
  void operator&(const LogLine &) {}
};
}  // namespace oracle_ref_stub
#define ORACLE_STUB_SEV_INFO false
#define ORACLE_STUB_SEV_WARNING false
#define ORACLE_STUB_SEV_ERROR false
#define ORACLE_STUB_SEV_FATAL true
#define LOG(sev) ::oracle_ref_stub::LogLine(ORACLE_STUB_SEV_##sev)
#define VLOG(n) ::oracle_ref_stub::LogLine(false)
#define CHECK(cond) (cond) ? (void)0 : ::oracle_ref_stub::Voidify() & ::oracle_ref_stub::LogLine(true) << "CHECK failed: " #cond " "
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#define CHECK_LT(a, b) CHECK((a) < (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#endif
