// tests/emu/hip/hip_runtime.h — placeholder so `#include <hip/hip_runtime.h>` in kernel headers resolves under the host emulator
