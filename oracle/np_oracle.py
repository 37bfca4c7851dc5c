"""numpy restatement of the reference's embedding extractors (TEST INFRASTRUCTURE - see
oracle/__init__.py).

Activations are held as [T, C] (frames x channels), i.e. the Kaldi matrix layout; the
reference holds them as [1, C, T] only because torch conv1d wants channels first
(framework.py:29-31).  Every function cites the reference lines it restates (paths are
relative to /root/reference/pytorch).  `dtype` defaults to float32 like the reference;
float64 gives a "ground truth" to judge both the reference's and the HIP path's rounding.

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which were produced by
running the reference itself (oracle/gen_golden.py).
"""

import numpy as np


# ----------------------------------------------------------------------------- layers

def tdnn_affine(x, weight, bias, context, pad=True):
    """libs/nnet/components.py:107-149 (ctor 30-97).

    x [T, Cin]; weight [Cout, Cin, tot_context] (dense kernel as stored in the checkpoint;
    taps outside `context` are masked out, components.py:78-83,133-136); bias [Cout]|None.
    Zero padding of -left / +right frames (components.py:51-52,116-117) => T_out == T_in.
    Cross-correlation, kernel index k <-> offset left_context + k.
    """
    left = context[0] if context[0] < 0 else 0
    right = context[-1] if context[-1] > 0 else 0
    tot = right - left + 1
    assert weight.shape[2] == tot, (weight.shape, context)
    T = x.shape[0]
    if pad:
        xp = np.zeros((T - left + right, x.shape[1]), dtype=x.dtype)
        xp[-left:-left + T] = x
    else:
        xp = x
    assert xp.shape[0] >= tot                       # components.py:119
    T_out = xp.shape[0] - tot + 1
    y = np.zeros((T_out, weight.shape[0]), dtype=x.dtype)
    for off in context:                             # active taps only
        k = off - left
        y += xp[k:k + T_out] @ weight[:, :, k].T.astype(x.dtype)
    if bias is not None:
        y += bias.astype(x.dtype)
    return y


def batchnorm_eval(x, sd, prefix, eps=1e-5):
    """torch BatchNorm1d/2d in eval mode on the channel (last) axis:
    (x - running_mean) / sqrt(running_var + eps) * weight + bias (components.py:374)."""
    dt = x.dtype
    rm = sd[prefix + ".running_mean"].astype(dt)
    rv = sd[prefix + ".running_var"].astype(dt)
    y = (x - rm) / np.sqrt(rv + dt.type(eps))
    if prefix + ".weight" in sd:                    # affine=True
        y = y * sd[prefix + ".weight"].astype(dt) + sd[prefix + ".bias"].astype(dt)
    return y


def _act(x, name):
    """libs/nnet/activation.py:81-106 (only what the target models use)."""
    if name in ("", None, False):
        return x
    if name == "relu":
        return np.maximum(x, 0)
    if name == "tanh":
        return np.tanh(x)
    if name == "sigmoid":
        return 1.0 / (1.0 + np.exp(-x))
    raise ValueError("oracle: nonlinearity %r not restated" % (name,))


def relu_bn_tdnn(x, sd, prefix, context=(0,), nonlinearity="relu", bn=True, bn_relu=False):
    """ReluBatchNormTdnnLayer: affine -> ReLU -> BN (components.py:418-431,434-461);
    order flips when "bn-relu" is set (components.py:365-386)."""
    w = sd[prefix + ".affine.weight"]
    b = sd.get(prefix + ".affine.bias")
    y = tdnn_affine(x, w, b, list(context))
    if bn_relu:
        if bn:
            y = batchnorm_eval(y, sd, prefix + ".batchnorm")
        y = _act(y, nonlinearity)
    else:
        y = _act(y, nonlinearity)
        if bn:
            y = batchnorm_eval(y, sd, prefix + ".batchnorm")
    return y


def statistics_pooling(x, stddev=True, unbiased=False, eps=1.0e-10):
    """libs/nnet/pooling.py:58-67: mean over frames, two-pass biased variance,
    std = sqrt(clamp(var, eps)); returns [2C] (mean || std)."""
    T = x.shape[0]
    mean = x.mean(axis=0, dtype=x.dtype)
    if not stddev:
        return mean
    counts = T - 1 if (unbiased and T > 1) else T
    var = ((x - mean) ** 2).sum(axis=0, dtype=x.dtype) / x.dtype.type(counts)
    std = np.sqrt(np.maximum(var, x.dtype.type(eps)))
    return np.concatenate([mean, std])


# ----------------------------------------------------------------------------- x-vector

XVECTOR_CONTEXTS = {                                # model/xvector.py:28-32
    "tdnn1": (-2, -1, 0, 1, 2), "tdnn2": (-2, 0, 2), "tdnn3": (-3, 0, 3),
    "tdnn4": (0,), "tdnn5": (0,),
}


def xvector_embed(x, sd, position="far", taps=None):
    """model/xvector.py:84-98.  x [T, D] -> [512].  `taps` may collect intermediates."""
    for name in ("tdnn1", "tdnn2", "tdnn3", "tdnn4", "tdnn5"):
        x = relu_bn_tdnn(x, sd, name, XVECTOR_CONTEXTS[name])
        if taps is not None:
            taps[name] = x
    s = statistics_pooling(x)[None, :]              # [1, 3000]: one pooled "frame"
    if taps is not None:
        taps["stats"] = s[0]
    if position == "far":
        return tdnn_affine(s, sd["tdnn6.affine.weight"], sd["tdnn6.affine.bias"], [0])[0]
    if position == "near":
        h = relu_bn_tdnn(s, sd, "tdnn6")
        return tdnn_affine(h, sd["tdnn7.affine.weight"], sd["tdnn7.affine.bias"], [0])[0]
    raise TypeError(position)


# ----------------------------------------------------------------------------- ECAPA

def conv1d_1x1(x, sd, prefix):
    """nn.Conv1d(kernel_size=1) on [T|1, Cin] (ecapa_tdnn_xvector.py:102,105,165,169)."""
    w = sd[prefix + ".weight"][:, :, 0].astype(x.dtype)
    return x @ w.T + sd[prefix + ".bias"].astype(x.dtype)


def res2net_block(x, sd, prefix, scale=8, dilation=1):
    """ecapa_tdnn_xvector.py:61-75 (ctor 42-59): split C into `scale` groups; group 0
    passes through; group i>=1: sp = (sp_prev + x_i if i>=2 else x_i) -> TDNN[-d,0,d]+ReLU+BN."""
    C = x.shape[1]
    w = C // scale
    ctx = (-dilation, 0, dilation)
    ys = [x[:, :w]]
    sp = None
    for i in range(scale - 1):
        xi = x[:, (i + 1) * w:(i + 2) * w]
        sp = xi if i == 0 else sp + xi
        sp = relu_bn_tdnn(sp, sd, "%s.blocks.%d" % (prefix, i), ctx)
        ys.append(sp)
    return np.concatenate(ys, axis=1)


def se_connect(x, sd, prefix):
    """ecapa_tdnn_xvector.py:97-111: time-mean -> 1x1 conv -> ReLU -> 1x1 conv -> sigmoid -> scale."""
    m = x.mean(axis=0, dtype=x.dtype)[None, :]
    h = np.maximum(conv1d_1x1(m, sd, prefix + ".se.1"), 0)
    s = 1.0 / (1.0 + np.exp(-conv1d_1x1(h, sd, prefix + ".se.3")))
    return x * s.astype(x.dtype)


def se_res2block(x, sd, prefix, dilation):
    """ecapa_tdnn_xvector.py:139-149 (ctor 119-137); shortcut conv only if Cin != Cout."""
    residual = x
    if prefix + ".shortcut.weight" in sd:
        residual = conv1d_1x1(x, sd, prefix + ".shortcut")
    y = relu_bn_tdnn(x, sd, prefix + ".conv_relu_bn1")
    y = res2net_block(y, sd, prefix + ".res2net_block", 8, dilation)
    y = relu_bn_tdnn(y, sd, prefix + ".conv_relu_bn2")
    y = se_connect(y, sd, prefix + ".se")
    return y + residual


def attentive_stats_pool(x, sd, prefix, time_attention=True):
    """ecapa_tdnn_xvector.py:173-188 (ctor 157-171).  Global std uses torch.var's default
    UNBIASED estimate + 1e-5 inside the sqrt (177-178); attention = conv -> ReLU -> BN ->
    tanh -> conv -> softmax over time; weighted std = sqrt(clamp(E[x^2]-mean^2, 1e-5))."""
    dt = x.dtype
    T = x.shape[0]
    if time_attention:
        gm = x.mean(axis=0, dtype=dt)
        gv = ((x - gm) ** 2).sum(axis=0, dtype=dt) / dt.type(T - 1)
        gs = np.sqrt(gv + dt.type(1e-5))
        x_in = np.concatenate([x, np.broadcast_to(gm, x.shape), np.broadcast_to(gs, x.shape)], axis=1)
    else:
        x_in = x
    h = np.maximum(conv1d_1x1(x_in, sd, prefix + ".attention.0"), 0)
    h = np.tanh(batchnorm_eval(h, sd, prefix + ".attention.2"))
    e = conv1d_1x1(h, sd, prefix + ".attention.4")
    e = e - e.max(axis=0, keepdims=True)
    a = np.exp(e)
    a = a / a.sum(axis=0, keepdims=True, dtype=dt)
    mean = (a * x).sum(axis=0, dtype=dt)
    resid = (a * x * x).sum(axis=0, dtype=dt) - mean * mean
    std = np.sqrt(np.maximum(resid, dt.type(1e-5)))
    return np.concatenate([mean, std])


def ecapa_embed(x, sd, position="near", fc2_nonlinearity="relu", taps=None):
    """model/ecapa_tdnn_xvector.py:403-426 (ctor 201-357), pooling="ecpa-attentive",
    fc1 optional (present iff 'fc1.affine.weight' in sd)."""
    x = relu_bn_tdnn(x, sd, "layer1", (-2, -1, 0, 1, 2))
    x1 = se_res2block(x, sd, "layer2", 2)
    x2 = se_res2block(x + x1, sd, "layer3", 3)
    x3 = se_res2block(x + x1 + x2, sd, "layer4", 4)
    if taps is not None:
        taps.update(layer1=x, x1=x1, x2=x2, x3=x3)
    y = relu_bn_tdnn(np.concatenate([x1, x2, x3], axis=1), sd, "mfa")
    s = attentive_stats_pool(y, sd, "stats")
    if taps is not None:
        taps.update(mfa=y, stats=s)
    s = batchnorm_eval(s[None, :], sd, "bn_stats")
    has_fc1 = "fc1.affine.weight" in sd
    if position == "far":
        assert has_fc1
        return tdnn_affine(s, sd["fc1.affine.weight"], sd["fc1.affine.bias"], [0])[0]
    if has_fc1:
        s = relu_bn_tdnn(s, sd, "fc1")
    if position == "near_affine":
        return tdnn_affine(s, sd["fc2.affine.weight"], sd["fc2.affine.bias"], [0])[0]
    if position == "near":
        return relu_bn_tdnn(s, sd, "fc2", nonlinearity=fc2_nonlinearity)[0]
    raise TypeError(position)


# ----------------------------------------------------------------------------- wrapper

def chunk_plan(num_frames, max_chunk=10000):
    """libs/nnet/framework.py:34-47: ceil(T/maxChunk) near-equal chunks, the last one
    takes the remainder.  Returns [(offset, length), ...]."""
    num_split = (num_frames + max_chunk - 1) // max_chunk
    split = num_frames // num_split
    plan = [(i * split, split) for i in range(num_split - 1)]
    off = (num_split - 1) * split
    plan.append((off, num_frames - off))
    return plan


def extract_embedding(embed_fn, feats, max_chunk=10000, dtype=np.float32):
    """libs/nnet/framework.py:18-52 (`for_extract_embedding`): per-chunk embeddings,
    frame-weighted mean, in the reference's operation order."""
    feats = np.asarray(feats, dtype=dtype)
    T = feats.shape[0]
    plan = chunk_plan(T, max_chunk)
    stats = 0.0
    for off, n in plan[:-1]:
        stats = stats + dtype(n) * embed_fn(feats[off:off + n])
    off, n = plan[-1]
    last = embed_fn(feats[off:off + n])
    return ((stats + dtype(n) * last) / dtype(T)).astype(dtype)


def cast_state_dict(sd, dtype):
    return {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in sd.items()}


# ----------------------------------------------------------------------------- ResNet (2-D)

def conv2d(x, weight, stride=1, padding=1):
    """nn.Conv2d(bias=False) as used by libs/nnet/resnet.py:12-20.  x [T, F, Cin] (time, frequency,
    channels - the reference holds [B, C, F, T]); weight [Cout, Cin, kF, kT] exactly as in the
    checkpoint (kernel height = frequency axis, width = time axis)."""
    T, F, Cin = x.shape
    Cout, _, kF, kT = weight.shape
    xp = np.zeros((T + 2 * padding, F + 2 * padding, Cin), dtype=x.dtype)
    xp[padding:padding + T, padding:padding + F] = x
    To = (T + 2 * padding - kT) // stride + 1
    Fo = (F + 2 * padding - kF) // stride + 1
    y = np.zeros((To, Fo, Cout), dtype=x.dtype)
    for df in range(kF):
        for dt in range(kT):
            patch = xp[dt:dt + (To - 1) * stride + 1:stride, df:df + (Fo - 1) * stride + 1:stride]
            y += patch @ weight[:, :, df, dt].T.astype(x.dtype)
    return y


def se_block_2d(x, sd, prefix):
    """components.py:622-639: global average over (F, T) -> Linear -> ReLU -> Linear -> Sigmoid -> scale."""
    m = x.mean(axis=(0, 1), dtype=x.dtype)
    h = np.maximum(m @ sd[prefix + ".fc_1.weight"].T.astype(x.dtype) + sd[prefix + ".fc_1.bias"].astype(x.dtype), 0)
    s = 1.0 / (1.0 + np.exp(-(h @ sd[prefix + ".fc_2.weight"].T.astype(x.dtype) + sd[prefix + ".fc_2.bias"].astype(x.dtype))))
    return x * s.astype(x.dtype)


def basic_block_preact(x, sd, prefix, stride):
    """BasicBlock full pre-activation form (resnet.py:87-104): BN-ReLU-conv, BN-ReLU-conv, SE, + identity (no final ReLU);
    the down-sampling branch works on the raw block input."""
    identity = x
    y = np.maximum(batchnorm_eval(x, sd, prefix + ".bn1"), 0)
    y = conv2d(y, sd[prefix + ".conv1.weight"], stride, 1)
    y = np.maximum(batchnorm_eval(y, sd, prefix + ".bn2"), 0)
    y = conv2d(y, sd[prefix + ".conv2.weight"], 1, 1)
    if prefix + ".se.fc_1.weight" in sd:
        y = se_block_2d(y, sd, prefix + ".se")
    if prefix + ".downsample.0.weight" in sd:
        identity = batchnorm_eval(conv2d(x, sd[prefix + ".downsample.0.weight"], stride, 0), sd, prefix + ".downsample.1")
    return y + identity


def basic_block(x, sd, prefix, stride):
    """BasicBlock original form (resnet.py:70-85): conv-BN-ReLU-conv-BN-SE-(+identity)-ReLU;
    downsample = 1x1 conv (stride) + BN when present in the checkpoint."""
    identity = x
    y = conv2d(x, sd[prefix + ".conv1.weight"], stride, 1)
    y = np.maximum(batchnorm_eval(y, sd, prefix + ".bn1"), 0)
    y = conv2d(y, sd[prefix + ".conv2.weight"], 1, 1)
    y = batchnorm_eval(y, sd, prefix + ".bn2")
    if prefix + ".se.fc_1.weight" in sd:
        y = se_block_2d(y, sd, prefix + ".se")
    if prefix + ".downsample.0.weight" in sd:
        identity = batchnorm_eval(conv2d(x, sd[prefix + ".downsample.0.weight"], stride, 0), sd, prefix + ".downsample.1")
    return np.maximum(y + identity, 0)


def resnet_trunk(x, sd, prefix="resnet", layers=(3, 4, 6, 3), preact=False):
    """ResNet._forward_impl (resnet.py:352-368) with head_conv 3x3/stride 1, no max-pool."""
    y = conv2d(x, sd[prefix + ".conv1.weight"], 1, 1)
    y = np.maximum(batchnorm_eval(y, sd, prefix + ".bn1"), 0)
    block = basic_block_preact if preact else basic_block
    for li, n in enumerate(layers):
        for b in range(n):
            y = block(y, sd, "%s.layer%d.%d" % (prefix, li + 1, b), 2 if (li > 0 and b == 0) else 1)
    return y


def input_sequence_norm(x, mean_norm=True, std_norm=False, eps=1e-10):
    """components.py:780-842: per-utterance (x - mean_t) / max(std_t, eps) with torch.std's unbiased estimate."""
    mean = x.mean(axis=0, dtype=x.dtype) if mean_norm else 0.0
    if std_norm:
        std = np.maximum(x.std(axis=0, ddof=1, dtype=x.dtype), x.dtype.type(eps))
    else:
        std = 1.0
    return ((x - mean) / std).astype(x.dtype)


def resnet_embed(x, sd, position="near", fc2_nonlinearity="relu", layers=(3, 4, 6, 3), cmvn=None, preact=False):
    """model/resnet_xvector.py:183-208.  x [T, D] -> trunk on [T, F=D, 1] -> [T', F', C] ->
    reshape to channel index c*F' + f (resnet_xvector.py:193) -> StatisticsPooling -> fc2."""
    if cmvn is not None:
        x = input_sequence_norm(x, **cmvn)
    y = resnet_trunk(x[:, :, None], sd, "resnet", layers, preact)   # [T', F', C]
    To, Fo, C = y.shape
    feat = y.transpose(0, 2, 1).reshape(To, C * Fo)                 # column index c*F' + f
    s = statistics_pooling(feat)[None, :]
    has_fc1 = "fc1.affine.weight" in sd
    if position == "far":
        assert has_fc1
        return tdnn_affine(s, sd["fc1.affine.weight"], sd["fc1.affine.bias"], [0])[0]
    if has_fc1:
        s = relu_bn_tdnn(s, sd, "fc1")
    if position == "near_affine":
        return tdnn_affine(s, sd["fc2.affine.weight"], sd["fc2.affine.bias"], [0])[0]
    if position == "near":
        return relu_bn_tdnn(s, sd, "fc2", nonlinearity=fc2_nonlinearity)[0]
    raise TypeError(position)
