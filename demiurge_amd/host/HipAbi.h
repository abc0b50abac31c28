// The adapters are compiled against include/demiurge_hip.h and run against whatever libdemiurge_hip.so is installed: they refuse a
// library whose DMG_VERSION differs (argument lists and struct sizes change between versions), with the reason in error().
#pragma once
#include <string>
#include "../../include/demiurge_hip.h"

inline bool hip_abi_matches(std::string *why)
{
    const int have = dmg_version();
    if (have == DMG_VERSION) return true;
    if (why) *why = "libdemiurge_hip is version " + std::to_string(have) + ", this adapter was built against " + std::to_string(DMG_VERSION);
    return false;
}
