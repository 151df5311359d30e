#include "../../boost_stub.hpp"
