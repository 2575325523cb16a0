// impl_f32.hip -- fp32 instantiation (the reference's Float32; BASELINE config 4 mixed precision:
// fp32 storage and pair math, fp64 global energy/virial reduction).
#include "dd.hpp"

namespace emdee {
template struct Factory<float>;
}

#ifdef EMDEE_BOUNDS
namespace emdee {
void bounds_poll_f32(int out[3]) { bounds_poll_here(out); }
}  // namespace emdee
#endif
