// Shared between capi_mesh.cpp (host only) and gmg_*.hip (device): handle structs + error slot.
#pragma once
#include "../../include/te_hip.h"
#include "mesh.hpp"
#include <string>

struct te_mesh {
	te::Tree tree;
};
struct te_hier {
	te::Hierarchy h;
};

namespace te
{
std::string &lastError();
int          fail(int code, const std::string &msg);
} // namespace te
