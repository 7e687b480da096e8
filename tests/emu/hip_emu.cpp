// tests/emu/hip_emu.cpp — TEST INFRASTRUCTURE ONLY (see hip_emu.h)
#include "hip_emu.h"
namespace emu {
Block* B = nullptr;
void trampoline() { Block* b = B; b->body(); b->fibers[b->cur].done = true; }
}
alignas(16) double qm_smem[160 * 1024 / 8];
