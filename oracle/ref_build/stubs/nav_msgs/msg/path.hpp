#include "../../ros_stub.hpp"
