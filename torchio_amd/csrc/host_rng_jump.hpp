// Jump-ahead for the mt19937 state chain (host_rng_jump.cpp): declarations shared with host_rng.cpp.
#pragma once
#include <stdint.h>

#include <memory>
#include <vector>

namespace tio_host_rng {

// the characteristic polynomial could be computed and verified (first call: ~0.1 s, once per process)
bool jump_available();
// polynomials that carry a state t * segment_blocks twists ahead, t = 1 .. count (shared, immutable: concurrent plans may
// extend the cache behind them); empty when unavailable
typedef std::shared_ptr<const std::vector<uint64_t>> JumpPolynomial;
std::vector<JumpPolynomial> jump_polynomials(int64_t segment_blocks, int count);
// out[0 .. 624) = `in` carried ahead by g (see the .cpp for the 31 bits that are not part of the state)
void jump_state(const uint32_t* in, const std::vector<uint64_t>& g, uint32_t* out);

}  // namespace tio_host_rng
