// Thin C-ABI shim over the REFERENCE's own pcg32.h, compiled from where it lies under /root/reference
// (oracle/Makefile target `ref`, output oracle/_ref/libref_pcg32.so -- git-ignored, never committed).
// Used only by tests/test_oracle_ref.py to pin oracle/tcnn_oracle.cpp's RNG and parameter initialisation
// bit-for-bit against the reference.  TEST INFRASTRUCTURE ONLY.
#include <pcg32/pcg32.h>

#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

extern "C" {

// Trainer's seeding (trainer.h:52-55): std::seed_seq{seed} -> pcg32{seeds.front()}
void ref_trainer_rng(uint32_t seed, uint64_t* state_inc) {
	std::seed_seq seq{seed};
	std::vector<uint32_t> seeds(2);
	seq.generate(std::begin(seeds), std::end(seeds));
	tcnn::pcg32 rng{seeds.front()};
	state_inc[0] = rng.state;
	state_inc[1] = rng.inc;
}

// cpp_api.cu:133-136: pcg32 rng{seed}
void ref_module_rng(uint64_t seed, uint64_t* state_inc) {
	tcnn::pcg32 rng{seed};
	state_inc[0] = rng.state;
	state_inc[1] = rng.inc;
}

void ref_next_floats(uint64_t* state_inc, uint32_t n, float* out) {
	tcnn::pcg32 rng;
	rng.state = state_inc[0];
	rng.inc = state_inc[1];
	for (uint32_t i = 0; i < n; ++i) out[i] = rng.next_float();
	state_inc[0] = rng.state;
	state_inc[1] = rng.inc;
}

void ref_next_uints(uint64_t* state_inc, uint32_t n, uint32_t* out) {
	tcnn::pcg32 rng;
	rng.state = state_inc[0];
	rng.inc = state_inc[1];
	for (uint32_t i = 0; i < n; ++i) out[i] = rng.next_uint();
	state_inc[0] = rng.state;
	state_inc[1] = rng.inc;
}

void ref_advance(uint64_t* state_inc, int64_t delta) {
	tcnn::pcg32 rng;
	rng.state = state_inc[0];
	rng.inc = state_inc[1];
	rng.advance(delta);
	state_inc[0] = rng.state;
	state_inc[1] = rng.inc;
}

}
