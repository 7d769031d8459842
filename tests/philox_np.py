"""Philox4x32-10 in numpy (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123's philox4x32_R(10)),
written from the published round function, NOT from oracle/lg_oracle.c or the kernels -- it is the independent third party
between them: tests/test_philox.py checks it against Random123's known-answer vectors, tools/make_golden.py feeds the
reference's own reset / resample / push / noise code with uniforms drawn from it, and both the oracle and the HIP path must then
reproduce the reference's outputs from nothing but (seed; env, step, purpose, block).

Key order of the build (oracle/lg_oracle.c:rand4, csrc/lg_device.h): counter = (env, step, purpose, block), key = (seed lo, seed hi);
a uniform is the top 24 bits of a 32-bit output word (torch.rand's float32 resolution)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

# purposes (third counter word)
NOISE, CMD_STEP, CMD_RESET, DOF, ROOT, PUSH, TERRAIN, NOISE_H = range(8)


def philox4x32_10(counter, key):
    """counter [..., 4], key [..., 2] (broadcastable), any integer dtype -> uint32 [..., 4]."""
    c = np.asarray(counter).astype(np.uint64) & MASK
    k = np.asarray(key).astype(np.uint64) & MASK
    c0, c1, c2, c3 = (c[..., i] for i in range(4))
    k0, k1 = k[..., 0], k[..., 1]
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                      # 32 x 32 -> 64 bit products
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & MASK, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & MASK
        k0, k1 = (k0 + np.uint64(W0)) & MASK, (k1 + np.uint64(W1)) & MASK
    return np.stack(np.broadcast_arrays(c0, c1, c2, c3), axis=-1).astype(np.uint32)


def uniforms(seed, env, step, purpose, block):
    """float32 [..., 4] in [0, 1): the four uniforms of Philox block (seed; env, step, purpose, block); arguments broadcast."""
    env, step, purpose, block = np.broadcast_arrays(np.asarray(env, np.int64), np.asarray(step, np.int64), np.asarray(purpose, np.int64), np.asarray(block, np.int64))
    ctr = np.stack((env, step, purpose, block), axis=-1)
    seed = int(seed)
    out = philox4x32_10(ctr, np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint64))
    return ((out >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def lanes(seed, env_ids, step, purpose, first_lane, count):
    """float32 [len(env_ids), count]: consecutive uniforms ``first_lane .. first_lane + count - 1`` of a purpose's stream
    (lane l lives in block l // 4, word l % 4) for every env of ``env_ids``."""
    env_ids = np.asarray(env_ids, np.int64).reshape(-1, 1)
    l = np.arange(first_lane, first_lane + count, dtype=np.int64).reshape(1, -1)
    u = uniforms(seed, env_ids, step, purpose, l // 4)                 # [n, count, 4]
    return np.take_along_axis(u, (l % 4)[..., None].repeat(env_ids.shape[0], 0), axis=-1)[..., 0]


def observation_noise(seed, num_envs, step, num_obs, K, L):
    """float32 [num_envs, num_obs]: the uniform the build draws for observation element i.  The first 48 elements are four
    groups of (K limbs x L joints): element g*12 + k*L + j <- purpose NOISE, block (g*K + k)*2 + j // 4, word j % 4;
    height sample i <- purpose NOISE_H, lane i (oracle/lg_oracle.c:compute_observations_env)."""
    envs = np.arange(num_envs)
    out = np.zeros((num_envs, num_obs), np.float32)
    for g in range(4):
        for k in range(K):
            for j in range(L):
                out[:, g * 12 + k * L + j] = uniforms(seed, envs, step, NOISE, (g * K + k) * 2 + j // 4)[:, j % 4]
    if num_obs > 48:
        out[:, 48:] = lanes(seed, envs, step, NOISE_H, 0, num_obs - 48)
    return out
