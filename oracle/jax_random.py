"""jax.random's key handling restated in numpy (TEST INFRASTRUCTURE, like everything under oracle/): the pin of `kbj_config.command_mode = 2`.

The reference's in-tree samplers (`/root/reference/train.py:725-752, 782-785, 834-836`) call `jax.random.split / uniform / bernoulli / randint`.
jax is not in the image, so this file restates what those functions compute on a threefry2x32 key (jax 0.6.0, `requirements.lock:72`) from the
published algorithm, and `tests/test_oracle_task.py` checks it against jax.random's PUBLIC known answers:
  * `jax.random.split(PRNGKey(0))`  = [[4146024105, 967050713], [2718843009, 1272950319]]   (original, `jax_threefry_partitionable=False`)
  * `jax.random.split(PRNGKey(0))`  = [[1797259609, 2579123966], [928981903, 3453687069]]    (partitionable: the default since jax 0.5)
  * `jax.random.uniform(PRNGKey(0))` = 0.41845703                                             (original)
Both key-derivation modes are here; the product and the C++ oracle implement the PARTITIONABLE one (the reference's pinned jax uses it by default).
Everything below the call key is therefore defined against jax; which key each call receives is ksim's (un-vendored) business and stays this
build's own definition. Values a live JAX would produce for whole commands are UNVERIFIED until a JAX box writes fixtures."""
import numpy as np

M32 = 0xFFFFFFFF
_ROT = (13, 15, 26, 6, 17, 29, 16, 24)


def threefry2x32(key, c0: int, c1: int):
    """Threefry-2x32, 20 rounds (Salmon et al., SC'11), as jax's `threefry2x32_p` applies it to one counter pair."""
    k0, k1 = int(key[0]) & M32, int(key[1]) & M32
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0, x1 = (c0 + ks[0]) & M32, (c1 + ks[1]) & M32
    for g in range(5):
        for r in range(4):
            x0 = (x0 + x1) & M32
            rot = _ROT[(g & 1) * 4 + r]
            x1 = ((x1 << rot) | (x1 >> (32 - rot))) & M32
            x1 ^= x0
        x0 = (x0 + ks[(g + 1) % 3]) & M32
        x1 = (x1 + ks[(g + 2) % 3] + g + 1) & M32
    return x0, x1


def PRNGKey(seed: int):
    return (0, seed & M32)          # jax.random.PRNGKey(seed) for 0 <= seed < 2**32: [0, seed]


def split(key, num: int = 2, partitionable: bool = True):
    """jax.random.split: `num` new keys. partitionable: key_i = threefry(key, (0, i)); original: threefry over iota(2 num) taken in halves."""
    if partitionable:
        return [threefry2x32(key, 0, i) for i in range(num)]
    counts = list(range(2 * num))
    lo, hi = counts[:num], counts[num:]                   # threefry_2x32 splits the (even-length) counter array into its two halves
    outs = [threefry2x32(key, a, b) for a, b in zip(lo, hi)]
    flat = [o[0] for o in outs] + [o[1] for o in outs]     # ... and concatenates the two output halves
    return [(flat[2 * i], flat[2 * i + 1]) for i in range(num)]


def random_bits(key, n: int = 1, partitionable: bool = True):
    """jax.random.bits(key, (n,), uint32) (n = 1 also serves shape ())."""
    if partitionable:
        return [a ^ b for a, b in (threefry2x32(key, 0, i) for i in range(n))]
    m = n + (n & 1)
    counts = list(range(n)) + [0] * (m - n)                # an odd counter array is padded with a ZERO (threefry_2x32), the padded output dropped
    outs = [threefry2x32(key, a, b) for a, b in zip(counts[:m // 2], counts[m // 2:])]
    return ([o[0] for o in outs] + [o[1] for o in outs])[:n]


def _u01(bits: int) -> np.float32:
    return np.uint32((bits >> 9) | 0x3F800000).view(np.float32) - np.float32(1.0)


def uniform(key, n: int = 1, minval=0.0, maxval=1.0, partitionable: bool = True):
    """jax.random.uniform(key, (n,), float32, minval, maxval): mantissa fill in [1, 2) minus 1, scaled with a separate multiply and add, clamped at minval."""
    lo = np.broadcast_to(np.asarray(minval, np.float32), (n,))
    hi = np.broadcast_to(np.asarray(maxval, np.float32), (n,))
    u = np.array([_u01(b) for b in random_bits(key, n, partitionable)], np.float32)
    return np.maximum(lo, (u * (hi - lo)).astype(np.float32) + lo).astype(np.float32)


def bernoulli(key, p=0.5, n: int = 1, partitionable: bool = True):
    return uniform(key, n, partitionable=partitionable) < np.float32(p)


def randint(key, minval: int, maxval: int, partitionable: bool = True) -> int:
    """jax.random.randint(key, (), minval, maxval) for int32: two 32-bit draws from a split, combined modulo the span."""
    k1, k2 = split(key, 2, partitionable)
    hi, lo = random_bits(k1, 1, partitionable)[0], random_bits(k2, 1, partitionable)[0]
    span = (maxval - minval) & M32 if maxval > minval else 1
    mult = (2 ** 16) % span
    mult = (mult * mult) % span
    off = (((hi % span) * mult) & M32) + (lo % span)
    return minval + (off & M32) % span


def unified_command(key, ranges, arms_lo, arms_hi, partitionable: bool = True):
    """`UnifiedCommand.initial_command(..., rng=key)` as written in train.py:724-766, in jax.random's vocabulary. ranges: dict vx/vy/wz/bh/rx/ry -> (lo, hi)."""
    rng_a, rng_b, rng_c, rng_d, rng_e, rng_f, rng_g, rng_h, rng_i = split(key, 9, partitionable)
    vx = uniform(rng_b, 1, *ranges["vx"], partitionable=partitionable)
    vy = uniform(rng_c, 1, *ranges["vy"], partitionable=partitionable)
    wz = uniform(rng_d, 1, *ranges["wz"], partitionable=partitionable)
    bh = uniform(rng_e, 1, *ranges["bh"], partitionable=partitionable)
    rx = uniform(rng_f, 1, *ranges["rx"], partitionable=partitionable)
    ry = uniform(rng_g, 1, *ranges["ry"], partitionable=partitionable)
    arms = uniform(rng_h, 10, np.asarray(arms_lo, np.float32), np.asarray(arms_hi, np.float32), partitionable=partitionable)
    mask = bernoulli(rng_h, n=10, partitionable=partitionable)
    arms = arms * mask
    z, zz = np.zeros_like(vx), np.zeros_like(arms)
    modes = [np.concatenate(v) for v in ([vx, z, z, z, z, z, zz], [z, vy, z, z, z, z, zz], [z, z, wz, z, z, z, zz], [vx, vy, wz, z, z, z, arms],
                                         [z, z, z, bh, rx, ry, arms], [z, z, z, z, z, z, zz])]
    return modes[randint(rng_a, 0, 6, partitionable)].astype(np.float32)


def unified_command_call(key, prev, switch_prob, ranges, arms_lo, arms_hi, partitionable: bool = True):
    """`UnifiedCommand.__call__(prev_command, ..., rng=key)` (train.py:768-785)."""
    rng_a, rng_b = split(key, 2, partitionable)
    switch = bool(bernoulli(rng_a, switch_prob, 1, partitionable)[0])
    return unified_command(rng_b, ranges, arms_lo, arms_hi, partitionable) if switch else np.asarray(prev, np.float32)
