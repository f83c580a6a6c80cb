"""The seeded RNG path by which Plonky3's own Poseidon2 unit test builds its round constants, restated (test
infrastructure; none of these crates is in /root/reference -- they are published third-party code, pinned by the
reference's Cargo.toml to p3-* =0.4.3):

* `Xoroshiro128Plus::seed_from_u64(1)` (rand_xoshiro): the 128-bit state is filled from SplitMix64(seed); `next_u64` is
  s0 + s1 followed by the (24, 16, 37) xoroshiro update; `next_u32` is the HIGH half of `next_u64`.
* `StandardUniform` for `MontyField31` (p3-monty-31): `next_u32() >> 1`, rejected unless < P, taken as the MONTGOMERY
  word (`new_monty`).
* `Poseidon2::new_from_rng_128` (p3-poseidon2): 4 x [F; 16] initial external constants, 4 x [F; 16] terminal external
  constants, then 13 internal constants, drawn in that order.

Whether this restatement is right is not a matter of trust: tests/golden/poseidon2_p3_kat.json holds the input and the
expected output of Plonky3's `test_poseidon2_width_16_random`, and sixteen field elements only come out equal if the
RNG, the sampling, the draw order AND the permutation's layers are all the published ones."""
P = 2013265921
M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.x = seed & M64

    def next_u64(self):
        self.x = (self.x + 0x9E3779B97F4A7C15) & M64
        z = self.x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)


def _rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M64


class Xoroshiro128Plus:
    def __init__(self, seed):
        sm = SplitMix64(seed)
        self.s0, self.s1 = sm.next_u64(), sm.next_u64()

    def next_u64(self):
        s0, s1 = self.s0, self.s1
        r = (s0 + s1) & M64
        s1 ^= s0
        self.s0 = _rotl(s0, 24) ^ s1 ^ ((s1 << 16) & M64)
        self.s1 = _rotl(s1, 37)
        return r

    def next_u32(self):
        return self.next_u64() >> 32


def sample_monty31(rng):
    """One uniformly random BabyBear element; returned CANONICAL (the sampled word is the Montgomery form)."""
    r_inv = pow(1 << 32, P - 2, P)
    while True:
        v = rng.next_u32() >> 1
        if v < P:
            return v * r_inv % P


def poseidon2_constants_from_rng_128(seed):
    """(initial[64], terminal[64], internal[13]), canonical, as `Poseidon2BabyBear::<16>::new_from_rng_128` draws them."""
    rng = Xoroshiro128Plus(seed)
    initial = [sample_monty31(rng) for _ in range(64)]
    terminal = [sample_monty31(rng) for _ in range(64)]
    internal = [sample_monty31(rng) for _ in range(13)]
    return initial, terminal, internal
