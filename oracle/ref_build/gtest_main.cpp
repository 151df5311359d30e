// main() for the reference's own gtest file, run on the stub gtest (oracle/ref_build/stubs/gtest/gtest.h)
#include <gtest/gtest.h>
int main() { return ref_gtest::run_all(); }
