#include "../assimp_stub.hpp"
