// psk_tile_inst.hip -- one instantiation of the time-tiled front kernel per translation unit:
//   hipcc -DPSK_INST_S=8 -DPSK_INST_H=1 -c psk_tile_inst.hip -o psk_tile_S8_H1.o
#include "psk_tile_kernel.h"

#define PSK_CAT_(a, b, c, d) a##b##c##d
#define PSK_CAT(a, b, c, d) PSK_CAT_(a, b, c, d)

namespace psk {
hipError_t PSK_CAT(launch_tile_front_S, PSK_INST_S, _H, PSK_INST_H)(PSK_TILE_FRONT_ARGS)
{
    return launch_tile_front_inst<PSK_INST_S, PSK_INST_H>(plans, list, ch0, nch, max_tiles, states, rings, ring_cap, r_len, tiles, t_raw,
                                                          t_s, pf_chan, tile0, stream);
}
}  // namespace psk
