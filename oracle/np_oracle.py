"""Second, independent CPU restatement (numpy / torch-CPU autograd) -- TEST INFRASTRUCTURE ONLY.

Written separately from oracle/ppo_oracle.c so the two can be cross-checked before anything is
called golden.  Pure-Python loops: small cases only.
"""
import numpy as np


def compute_returns(rewards, terminal, discount):
    """src/collect_rollouts.jl:26-42 with a Float64 discount (running value Float64)."""
    r = np.asarray(rewards, np.float32)
    out = np.zeros_like(r)
    v = np.float32(0.0)
    for i in range(len(r) - 1, -1, -1):
        if terminal[i]:
            v = np.float32(0.0)
        v = np.float64(r[i]) + np.float64(discount) * np.float64(v)   # promotion to Float64
        out[i] = np.float32(v)
    return out


def compute_returns_f32(rewards, terminal, discount):
    r = np.asarray(rewards, np.float32)
    g = np.float32(discount)
    out = np.zeros_like(r)
    v = np.float32(0.0)
    for i in range(len(r) - 1, -1, -1):
        if terminal[i]:
            v = np.float32(0.0)
        v = np.float32(r[i] + np.float32(g * v))
        out[i] = v
    return out


def philox4x32_10(ctr, key):
    c = [int(x) for x in ctr]
    k = [int(x) for x in key]
    M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k[1]) & MASK, p0 & MASK]
        k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
    return np.array(c, np.uint32)


def index_to_action(index, actions_per_edge=4, edges=4):
    """test/quad_game_utilities.jl:95-105 (1-based); edges=3 is the tutorial notebook's triangle variant."""
    apq = edges * actions_per_edge
    quad = (index - 1) // apq + 1
    qa = (index - 1) % apq
    return quad, qa // actions_per_edge + 1, qa % actions_per_edge + 1


def action_mask(active_quad, actions_per_edge=4):
    """test/quad_game_utilities.jl:39-44."""
    apq = 4 * actions_per_edge
    req = np.repeat(~np.asarray(active_quad, bool), apq)
    return np.where(req, -np.inf, 0.0).astype(np.float32)


def unpack_params(params, F, HID, n_hidden=2):
    p = np.asarray(params, np.float32)
    out, o = [], 0
    dims = [(HID, F)] + [(HID, HID)] * (n_hidden - 1) + [(4, HID)]
    for (no, ni) in dims:
        W = p[o:o + no * ni].reshape((no, ni), order="F")
        o += no * ni
        b = p[o:o + no]
        o += no
        out.append((W, b))
    assert o == p.size
    return out


def mlp_logits(params, F, HID, x, n_hidden=2, dtype=np.float64):
    """x: [H,F] ints.  Returns logits [H*4] with type fastest (vec of the [4,H] output)."""
    layers = unpack_params(params, F, HID, n_hidden)
    a = np.asarray(x, dtype).T                          # [F,H]
    for (W, b) in layers[:-1]:
        z = W.astype(dtype) @ a + b.astype(dtype)[:, None]
        a = np.where(z > 0, z, dtype(0.01) * z)
    W, b = layers[-1]
    y = W.astype(dtype) @ a + b.astype(dtype)[:, None]   # [4,H]
    return y.T.reshape(-1)                               # column-major vec == (h, type)


def masked_softmax(logits, mask):
    l = np.asarray(logits, np.float64) + np.asarray(mask, np.float64)
    m = np.max(l)
    e = np.exp(l - m)
    return e / e.sum()


def categorical_sample(p, u):
    """Sequential inverse CDF in float32 (Distributions.jl semantics)."""
    p = np.asarray(p, np.float32)
    cp = np.float32(p[0])
    i = 0
    while cp <= np.float32(u) and i < len(p) - 1:
        i += 1
        cp = np.float32(cp + p[i])
    return i


def step_batch_grad_torch(params, F, HID, states, masks, actions0, p_old, adv, eps, entropy_weight, n_hidden=2):
    """step_batch! loss (src/train.jl:35-46,54-84) differentiated by torch autograd in float64."""
    import torch
    layers = unpack_params(params, F, HID, n_hidden)
    tl = [(torch.tensor(W, dtype=torch.float64, requires_grad=True),
           torch.tensor(b, dtype=torch.float64, requires_grad=True)) for (W, b) in layers]
    x = torch.tensor(np.asarray(states), dtype=torch.float64)          # [B,H,F]
    B, H, _ = x.shape
    a = x
    for (W, b) in tl[:-1]:
        a = torch.nn.functional.leaky_relu(a @ W.T + b, 0.01)
    W, b = tl[-1]
    y = a @ W.T + b                                                    # [B,H,4]
    logits = y.reshape(B, H * 4) + torch.tensor(np.asarray(masks), dtype=torch.float64)
    probs = torch.softmax(logits, dim=1)                               # [B,A]
    A = H * 4
    sel = probs[torch.arange(B), torch.tensor(np.asarray(actions0), dtype=torch.long)]
    advt = torch.tensor(np.asarray(adv), dtype=torch.float64)
    pot = torch.tensor(np.asarray(p_old), dtype=torch.float64)
    gain = sel / pot * advt
    clip = torch.where(advt >= 0, (1.0 + eps) * advt, (1.0 - eps) * advt)
    ppoloss = -torch.mean(torch.minimum(gain, clip))
    s = float(np.float32(1e-8))
    sp = (1.0 - s) * probs + s / A
    ent = torch.mean(-(sp * torch.log(sp)).sum(dim=1))
    entloss = -ent * entropy_weight
    (ppoloss + entloss).backward()
    g = []
    for (W, b) in tl:
        g.append(W.grad.numpy().ravel(order="F"))
        g.append(b.grad.numpy())
    return np.concatenate(g), float(ppoloss.detach()), float(entloss.detach())


def adam_step(params, grad, m, v, beta_pow, eta=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """Flux legacy Adam; returns new (params, m, v, beta_pow) without mutating inputs."""
    g = np.asarray(grad, np.float32).astype(np.float64)
    m2 = (beta1 * m.astype(np.float64) + (1 - beta1) * g).astype(np.float32)
    v2 = (beta2 * v.astype(np.float64) + ((1 - beta2) * g) * g).astype(np.float32)
    delta = (m2.astype(np.float64) / (1 - beta_pow[0]) /
             (np.sqrt(v2.astype(np.float64) / (1 - beta_pow[1])) + eps) * eta).astype(np.float32)
    return (params - delta).astype(np.float32), m2, v2, np.array([beta_pow[0] * beta1, beta_pow[1] * beta2])


# ---------------------------------------------------------------- bf16 compute mode (ppo_policy_set_dtype(PPO_DTYPE_BF16))
# The reference is Float32 only; this restates the build's bf16 mode (csrc/ppo_policy_bf16.hip): weights and layer
# inputs rounded to bfloat16 (round-to-nearest-even), exact products, accumulation here in float64 (the device
# accumulates in fp32 in MFMA order -> compare with a tolerance), bias/leakyrelu/softmax/loss in higher precision,
# backward signals dY, dZ2, dZ1 rounded to bf16 before they enter a product.  PARITY UNPINNED against the reference.
def bf16_round(x):
    """float -> nearest bfloat16 value (RNE), returned as float64 array."""
    a = np.ascontiguousarray(np.asarray(x, np.float32))
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).astype(np.float64)
    return out.reshape(a.shape)


def _lrelu(z):
    return np.where(z > 0, z, 0.01 * z)


def mlp_forward_bf16(params, F, HID, x):
    """x: [B,H,F] ints.  Returns (logits [B, H*4] float64, h1b, h2b) with the bf16 rounding points of the device."""
    (W1, b1), (W2, b2), (W3, b3) = unpack_params(params, F, HID, 2)
    W1b, W2b, W3b = bf16_round(W1), bf16_round(W2), bf16_round(W3)
    X = np.asarray(x, np.float64)                                        # int8 values: exact in bf16
    h1 = _lrelu(np.float32(X @ W1b.T + b1.astype(np.float64)))           # fp32 accumulator, fp32 leakyrelu
    h1b = bf16_round(h1)
    h2 = _lrelu(np.float32(h1b @ W2b.T + b2.astype(np.float64)))
    h2b = bf16_round(h2)
    y = h2b @ W3b.T + b3.astype(np.float64)                              # [B,H,4]
    return y.reshape(y.shape[0], -1), h1b, h2b


def action_probabilities_bf16(params, F, HID, x, masks):
    logits, _, _ = mlp_forward_bf16(params, F, HID, x)
    l = logits + np.asarray(masks, np.float64)
    l = l - l.max(axis=1, keepdims=True)
    e = np.exp(l)
    return e / e.sum(axis=1, keepdims=True)


def step_batch_grad_bf16(params, F, HID, states, masks, actions0, p_old, adv, eps, entropy_weight):
    """Gradient of the step_batch! loss (src/train.jl:35-46,54-84) in the bf16 compute mode.
    Returns (flat grad in Flux order, ppoloss, entropy_weight*entropyloss)."""
    (W1, b1), (W2, b2), (W3, b3) = unpack_params(params, F, HID, 2)
    W2b, W3b = bf16_round(W2), bf16_round(W3)
    X = np.asarray(states, np.float64)
    B, H, _ = X.shape
    A = H * 4
    logits, h1b, h2b = mlp_forward_bf16(params, F, HID, X)
    l = logits + np.asarray(masks, np.float64)
    l = l - l.max(axis=1, keepdims=True)
    e = np.exp(l)
    p = e / e.sum(axis=1, keepdims=True)
    a0 = np.asarray(actions0, np.int64)
    advd, pod = np.asarray(adv, np.float64), np.asarray(p_old, np.float64)
    psel = p[np.arange(B), a0]
    gain = psel / pod * advd
    clip = np.where(advd >= 0, (1.0 + eps) * advd, (1.0 - eps) * advd)
    unclipped = gain < clip
    ppoloss = -np.mean(np.where(unclipped, gain, clip))
    sA = float(np.float32(1e-8)) / A
    sp = p + sA
    lg = np.log(sp)
    entloss = entropy_weight * np.mean((sp * lg).sum(axis=1))            # entropy_weight * (-H)
    dp = (entropy_weight / B) * (lg + 1.0)
    dp[np.arange(B), a0] += np.where(unclipped, -(advd / pod) / B, 0.0)
    dl = p * (dp - (p * dp).sum(axis=1, keepdims=True))                  # [B,A]
    dY = bf16_round(dl).reshape(B, H, 4)                                 # device: rounded once in the forward kernel
    dW3 = np.einsum("bho,bhf->of", dY, h2b)
    db3 = dY.sum(axis=(0, 1))
    dZ2 = bf16_round(np.float32((dY @ W3b) * np.where(h2b > 0, 1.0, 0.01)))
    dW2 = np.einsum("bhf,bhk->fk", dZ2, h1b)
    db2 = dZ2.sum(axis=(0, 1))
    dZ1 = bf16_round(np.float32((dZ2 @ W2b) * np.where(h1b > 0, 1.0, 0.01)))
    dW1 = np.einsum("bhk,bhi->ki", dZ1, X)
    db1 = dZ1.sum(axis=(0, 1))
    g = np.concatenate([dW1.ravel(order="F"), db1, dW2.ravel(order="F"), db2, dW3.ravel(order="F"), db3])
    return g, float(ppoloss), float(entloss)
