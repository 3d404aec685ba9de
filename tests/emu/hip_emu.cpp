// Out-of-line part of the SIMT emulator (scheduler, fibers, worker pool). Test infrastructure only.
#define HIP_EMU_IMPLEMENTATION
#include "hip_emu.h"
