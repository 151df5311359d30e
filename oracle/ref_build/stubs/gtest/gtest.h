// STUB of the googletest macros the reference's mesh_layers/test/inflation_layer_test.cpp uses, so that
// the reference's own test file compiles and runs unmodified against the reference's own layer code.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
namespace ref_gtest
{
struct Case { std::string name; std::function<void()> fn; };
inline std::vector<Case>& cases() { static std::vector<Case> c; return c; }
inline int& failures() { static int f = 0; return f; }
struct Reg { Reg(const char* n, std::function<void()> f) { cases().push_back({ n, std::move(f) }); } };
// EXPECT_FLOAT_EQ: within 4 ULPs, like googletest's FloatingPoint<float>::AlmostEquals
inline bool float_almost_eq(float a, float b)
{
  if (std::isnan(a) || std::isnan(b)) return false;
  auto biased = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u + 1 : u | 0x80000000u; };
  const uint32_t x = biased(a), y = biased(b);
  return (x > y ? x - y : y - x) <= 4;
}
inline void report(bool ok, const char* expr, const char* file, int line)
{
  if (!ok) { ++failures(); std::printf("%s:%d: Failure: %s\n", file, line, expr); }
}
inline int run_all()
{
  for (auto& c : cases()) {
    const int before = failures();
    std::printf("[ RUN      ] %s\n", c.name.c_str());
    c.fn();
    std::printf(failures() == before ? "[       OK ] %s\n" : "[  FAILED  ] %s\n", c.name.c_str());
  }
  std::printf("[==========] %zu tests, %d failures\n", cases().size(), failures());
  return failures() ? 1 : 0;
}
}  // namespace ref_gtest
#define TEST(suite, name) \
  static void suite##_##name##_body(); \
  static ::ref_gtest::Reg suite##_##name##_reg(#suite "." #name, suite##_##name##_body); \
  static void suite##_##name##_body()
#define EXPECT_TRUE(x) ::ref_gtest::report(static_cast<bool>(x), "EXPECT_TRUE(" #x ")", __FILE__, __LINE__)
#define EXPECT_FALSE(x) ::ref_gtest::report(!static_cast<bool>(x), "EXPECT_FALSE(" #x ")", __FILE__, __LINE__)
#define EXPECT_FLOAT_EQ(a, b) ::ref_gtest::report(::ref_gtest::float_almost_eq(static_cast<float>(a), static_cast<float>(b)), "EXPECT_FLOAT_EQ(" #a ", " #b ")", __FILE__, __LINE__)
#define EXPECT_LT(a, b) ::ref_gtest::report((a) < (b), "EXPECT_LT(" #a ", " #b ")", __FILE__, __LINE__)
#define EXPECT_GT(a, b) ::ref_gtest::report((a) > (b), "EXPECT_GT(" #a ", " #b ")", __FILE__, __LINE__)
#define EXPECT_EQ(a, b) ::ref_gtest::report((a) == (b), "EXPECT_EQ(" #a ", " #b ")", __FILE__, __LINE__)
