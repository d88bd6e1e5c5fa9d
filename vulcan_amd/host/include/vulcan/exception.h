// exception.h — vulcan::Exception and the assertion macros of the reference
// (ref: include/vulcan/exception.h: what() = "file(line): text"; VULCAN_DEBUG*
// compile away under NDEBUG). Host only: device code lives behind the C ABI.
#pragma once

#include <exception>
#include <string>

namespace vulcan
{

class Exception : public std::exception
{
  public:

    Exception(int line, const std::string& file, const std::string& text) :
      line_(line), file_(file), text_(text),
      what_(file + "(" + std::to_string(line) + "): " + text)
    {
    }

    int line() const { return line_; }

    const std::string& file() const { return file_; }

    const std::string& text() const { return text_; }

    const char* what() const noexcept override { return what_.c_str(); }

  protected:

    int line_;

    std::string file_;

    std::string text_;

    std::string what_;
};

} // namespace vulcan

#define VULCAN_THROW(text) throw ::vulcan::Exception(__LINE__, __FILE__, text)

#define VULCAN_ASSERT_MSG(cond, text) do { if (!(cond)) VULCAN_THROW(text); } while (0)

#define VULCAN_ASSERT(cond) VULCAN_ASSERT_MSG(cond, "assertion failed: " #cond)

#ifdef NDEBUG
#define VULCAN_DEBUG_MSG(cond, text) do { } while (0)
#define VULCAN_DEBUG(cond) do { } while (0)
#else
#define VULCAN_DEBUG_MSG VULCAN_ASSERT_MSG
#define VULCAN_DEBUG VULCAN_ASSERT
#endif
