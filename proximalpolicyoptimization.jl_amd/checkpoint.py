"""Flux/BSON checkpoint interop for SimplePolicy.Policy.

The reference saves trained policies with `BSON.@save path policy` (examples/triangle/distance_weighted/
triangle_utilities.jl:370-373; fixtures test/output/*.bson): a document {policy: struct Main.SimplePolicy.Policy
[Chain(Dense(W,b,leakyrelu), ..., Dense(W,b,identity)), hidden_channels, num_hidden_layers], _backrefs: [...]} with
Float32 arrays stored column-major (`size` = Julia dims).  `save_policy` writes that document -- byte for byte what
BSON.jl wrote for the reference's own fixtures (tests/test_checkpoint.py) -- and `load_policy` reads it back into
the flat Flux.params vector the engine uses, so a policy trained by the reference runs on the HIP engine and
vice versa.  Only the BSON subset BSON.jl emits for this document is handled (disk.py codec); nothing in a file
is executed.
"""
import numpy as np

from .disk import _dec_doc, _enc_doc


def _dt(name, params=()):
    return {"tag": "datatype", "params": list(params), "name": list(name)}


_F32 = ("Core", "Float32")
_LRELU = ("NNlib", "#leakyrelu")
_IDENT = ("Main", "Base", "#identity")


def _array_doc(a, dims):
    return {"tag": "array", "type": _dt(_F32), "size": [int(d) for d in dims],
            "data": np.ascontiguousarray(a, "<f4").tobytes()}


def _dense_type(act, inline=False):
    # Dense{typeof(act), Matrix{Float32}, Vector{Float32}}: the two array types are back-references 9 and 10 in the
    # documents BSON.jl wrote for the reference's three-Dense-layer fixtures; `inline` spells them out instead
    if inline:
        return _dt(("Flux", "Dense"), [_dt(act), _dt(("Core", "Array"), [_dt(_F32), 2]), _dt(("Core", "Array"), [_dt(_F32), 1])])
    return _dt(("Flux", "Dense"), [_dt(act), {"tag": "backref", "ref": 9}, {"tag": "backref", "ref": 10}])


def layer_dims(in_channels, hidden_channels, num_hidden_layers, num_output):
    return [(hidden_channels, in_channels)] + [(hidden_channels, hidden_channels)] * (num_hidden_layers - 1) + \
        [(num_output, hidden_channels)]


def policy_document(params, in_channels, hidden_channels, num_hidden_layers, num_output):
    """The BSON.jl document of `BSON.@save path policy` for SimplePolicy.Policy (test/policy.jl:1-33).
    num_hidden_layers == 2 reproduces the reference's fixtures byte for byte (their `_backrefs` table included).  The
    reference holds no checkpoint of another depth, so for those the shared array datatypes are written inline and
    `_backrefs` is empty -- the same document without BSON.jl's back-reference compression (parity unpinned: nothing in
    the reference pins the bytes; the arrays, their order and the struct tags are the ones BSON.jl reads)."""
    inline = num_hidden_layers != 2
    p = np.ascontiguousarray(params, np.float32)
    dims = layer_dims(in_channels, hidden_channels, num_hidden_layers, num_output)
    if p.size != sum(o * i + o for (o, i) in dims):
        raise ValueError("parameter count does not match the layer sizes")
    acts = [_LRELU] * num_hidden_layers + [_IDENT]
    layers, off = [], 0
    for (o, i), act in zip(dims, acts):
        W = p[off:off + o * i]
        off += o * i
        b = p[off:off + o]
        off += o
        layers.append({"tag": "struct", "type": _dense_type(act, inline),
                       "data": [_array_doc(W, (o, i)), _array_doc(b, (o,)),
                                {"tag": "struct", "type": _dt(act), "data": []}]})
    chain = {"tag": "struct",
             "type": _dt(("Flux", "Chain"), [_dt(("Core", "Tuple"), [_dense_type(a, inline) for a in acts])]),
             "data": [{"tag": "tuple", "data": layers}]}
    backrefs = []
    for _ in range(0 if inline else 5):
        backrefs.append(_dt(("Core", "Array"), [_dt(_F32), 2]))
        backrefs.append(_dt(("Core", "Array"), [_dt(_F32), 1]))
    return {"policy": {"tag": "struct", "type": _dt(("Main", "SimplePolicy", "Policy")),
                       "data": [chain, int(hidden_channels), int(num_hidden_layers)]},
            "_backrefs": backrefs}


def encode_policy(params, in_channels, hidden_channels, num_hidden_layers, num_output):
    return _enc_doc(policy_document(params, in_channels, hidden_channels, num_hidden_layers, num_output))


def decode_policy(raw):
    """-> (flat Flux.params float32, in_channels, hidden_channels, num_hidden_layers, num_output).
    Raises ValueError for anything that is not a SimplePolicy.Policy of Float32 Dense layers with leakyrelu hidden
    activations and an identity output layer."""
    doc, _ = _dec_doc(memoryview(raw).tobytes(), 0)
    pol = doc.get("policy")
    if not isinstance(pol, dict) or pol.get("tag") != "struct" or pol["type"]["name"][-1] != "Policy":
        raise ValueError("not a `BSON.@save path policy` document of SimplePolicy.Policy")
    chain, hidden, nhl = pol["data"]
    if chain["type"]["name"] != ["Flux", "Chain"]:
        raise ValueError("policy.model is not a Flux.Chain")
    layers = chain["data"][0]["data"]
    flat, dims = [], []
    for k, L in enumerate(layers):
        if L["type"]["name"] != ["Flux", "Dense"]:
            raise ValueError("layer %d is not a Flux.Dense" % (k + 1))
        W, b, act = L["data"]
        want = "#identity" if k == len(layers) - 1 else "#leakyrelu"
        if act["type"]["name"][-1] != want:
            raise ValueError("layer %d activation is %s, expected %s" % (k + 1, act["type"]["name"][-1], want))
        for a in (W, b):
            if a.get("tag") != "array" or a["type"]["name"][-1] != "Float32":
                raise ValueError("layer %d holds a non-Float32 array" % (k + 1))
        o, i = [int(x) for x in W["size"]]
        if [int(x) for x in b["size"]] != [o]:
            raise ValueError("layer %d bias length does not match its weight" % (k + 1))
        flat.append(np.frombuffer(W["data"], "<f4"))          # column-major [out,in] == Flux.params order
        flat.append(np.frombuffer(b["data"], "<f4"))
        dims.append((o, i))
    if len(dims) != int(nhl) + 1 or any(d[0] != int(hidden) for d in dims[:-1]):
        raise ValueError("layer sizes do not match hidden_channels / num_hidden_layers")
    return np.concatenate(flat).astype(np.float32), dims[0][1], int(hidden), int(nhl), dims[-1][0]


def save_policy(path, policy):
    """BSON.@save path policy  for a HipPolicy (or anything with .params and the four constructor fields)."""
    raw = encode_policy(policy.params, policy.in_channels, policy.hidden_channels, policy.num_hidden_layers,
                        policy.num_output)
    with open(path, "wb") as f:
        f.write(raw)


def load_policy(path, policy_cls=None, **kw):
    """BSON.@load path policy.  With policy_cls (e.g. HipPolicy) returns a new policy holding the weights,
    otherwise the tuple of decode_policy."""
    with open(path, "rb") as f:
        params, fin, hid, nhl, out = decode_policy(f.read())
    if policy_cls is None:
        return params, fin, hid, nhl, out
    pol = policy_cls(fin, hid, nhl, out, **kw)
    pol.params = params
    return pol
