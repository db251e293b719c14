// psk_fast_inst.hip -- one instantiation of the wave-scan kernel per translation unit:
//   hipcc -DPSK_INST_S=8 -DPSK_INST_H=1 -DPSK_INST_E=0 -c psk_fast_inst.hip -o psk_fast_S8_H1_E0.o
// (S = samplesPerBaud, H = blocks of window history in registers, E = 0 screened / 1 exact timing)
#include "psk_fast_kernel.h"

#define PSK_CAT_(a, b, c, d, e, f) a##b##c##d##e##f
#define PSK_CAT(a, b, c, d, e, f) PSK_CAT_(a, b, c, d, e, f)

namespace psk {
hipError_t PSK_CAT(launch_fast_S, PSK_INST_S, _H, PSK_INST_H, _E, PSK_INST_E)(PSK_FAST_ARGS)
{
    return launch_fast_inst<PSK_INST_S, PSK_INST_H, (PSK_INST_E != 0)>(plans, list, ch0, nch, states, rings, ring_cap, yvs,
                                                                       fit_cap, y_len, r_len, stream);
}
}  // namespace psk
