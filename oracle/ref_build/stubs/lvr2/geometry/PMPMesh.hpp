#include "../../lvr2_stub.hpp"
