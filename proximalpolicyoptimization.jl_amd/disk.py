"""Out-of-core rollout store: mirror of DiskRollouts / DiskDataset (src/rollouts_to_disk.jl:1-171,
src/dataset.jl:1-82).

Two on-disk layouts live in the same directory:
  * `rollout.bin` -- the engine's streaming shard (C ABI: ppo_rollouts_attach_disk / ppo_rollouts_load_disk);
    written while the rollout is being collected (device -> pinned host -> writer thread).
  * the reference's own layout -- `trajectory.csv` + `states/sample_<k>.bson` (BSON.jl array documents) -- written
    by the host-side `update_` (like the reference, one sample at a time) or exported from a finished device
    rollout for small runs, so data interoperates with the reference's DiskDataset.
The BSON encoder below reproduces the reference's own fixture byte for byte (tests/golden/sample_1.bson).
"""
import csv
import os
import shutil
import struct

import numpy as np

_JULIA_TYPES = {np.dtype(np.int64): "Int64", np.dtype(np.int32): "Int32", np.dtype(np.int8): "Int8",
                np.dtype(np.uint8): "UInt8", np.dtype(np.uint32): "UInt32", np.dtype(np.float32): "Float32",
                np.dtype(np.float64): "Float64", np.dtype(np.bool_): "Bool"}
_NP_TYPES = {v: k for k, v in _JULIA_TYPES.items()}


# ------------------------------------------------------------------ minimal BSON (the subset BSON.jl emits for arrays)
def _cstr(s):
    return s.encode("utf-8") + b"\x00"


def _enc_value(key, v):
    k = _cstr(key)
    if isinstance(v, str):
        b = v.encode("utf-8") + b"\x00"
        return b"\x02" + k + struct.pack("<i", len(b)) + b
    if isinstance(v, dict):
        return b"\x03" + k + _enc_doc(v)
    if isinstance(v, (list, tuple)):
        return b"\x04" + k + _enc_doc({str(i): x for i, x in enumerate(v)})
    if isinstance(v, (bytes, bytearray)):
        return b"\x05" + k + struct.pack("<i", len(v)) + b"\x00" + bytes(v)
    if isinstance(v, (int, np.integer)):
        return b"\x12" + k + struct.pack("<q", int(v))
    if isinstance(v, (float, np.floating)):
        return b"\x01" + k + struct.pack("<d", float(v))
    raise TypeError("unsupported BSON value %r" % type(v))


def _enc_doc(d):
    body = b"".join(_enc_value(k, v) for k, v in d.items())
    return struct.pack("<i", len(body) + 5) + body + b"\x00"


def _dec_doc(buf, pos, as_list=False):
    n = struct.unpack_from("<i", buf, pos)[0]
    end = pos + n - 1
    pos += 4
    out = [] if as_list else {}
    while pos < end:
        t = buf[pos]
        pos += 1
        z = buf.index(b"\x00", pos)
        key = buf[pos:z].decode("utf-8")
        pos = z + 1
        if t == 0x02:
            ln = struct.unpack_from("<i", buf, pos)[0]
            v = buf[pos + 4:pos + 4 + ln - 1].decode("utf-8")
            pos += 4 + ln
        elif t in (0x03, 0x04):
            v, pos = _dec_doc(buf, pos, as_list=(t == 0x04))
        elif t == 0x05:
            ln = struct.unpack_from("<i", buf, pos)[0]
            v = bytes(buf[pos + 5:pos + 5 + ln])
            pos += 5 + ln
        elif t == 0x12:
            v = struct.unpack_from("<q", buf, pos)[0]
            pos += 8
        elif t == 0x10:
            v = struct.unpack_from("<i", buf, pos)[0]
            pos += 4
        elif t == 0x01:
            v = struct.unpack_from("<d", buf, pos)[0]
            pos += 8
        elif t == 0x08:
            v = bool(buf[pos])
            pos += 1
        else:
            raise ValueError("unsupported BSON element type 0x%02x" % t)
        if as_list:
            out.append(v)
        else:
            out[key] = v
    return out, end + 1


def bson_array_document(a):
    """BSON.jl document of an Array: size = Julia dims (column-major), data = raw little-endian bytes.
    A numpy C-order [d0,...,dk] array is the Julia array of dims (dk,...,d0) with the same memory."""
    a = np.ascontiguousarray(a)
    return {"tag": "array",
            "type": {"tag": "datatype", "params": [], "name": ["Core", _JULIA_TYPES[a.dtype]]},
            "size": [int(x) for x in a.shape[::-1]],
            "data": a.tobytes()}


def bson_encode_state(state):
    """BSON.@save path state  ->  {state: ...}.  Arrays become array documents; a StateData-like object becomes a
    dictionary of its two arrays (loads in Julia as a Dict; the reference's struct tag needs Julia to reproduce)."""
    if hasattr(state, "vertex_score"):
        mask = state.mask_vector() if hasattr(state, "mask_vector") else np.asarray(state.action_mask)
        payload = {"vertex_score": bson_array_document(np.asarray(state.vertex_score)),
                   "action_mask": bson_array_document(np.asarray(mask, np.float32))}
    else:
        payload = bson_array_document(np.asarray(state))
    return _enc_doc({"state": payload})


def _array_from_doc(d):
    dt = _NP_TYPES[d["type"]["name"][-1]]
    return np.frombuffer(d["data"], dt).reshape([int(x) for x in d["size"]][::-1]).copy()


def bson_decode_state(raw):
    doc, _ = _dec_doc(memoryview(raw).tobytes(), 0)
    st = doc["state"]
    if st.get("tag") == "array":
        return _array_from_doc(st)
    return {k: _array_from_doc(v) for k, v in st.items()}


def _fmt32(x):
    """Julia prints a Float32 with the shortest round-trip representation (0.5, 6.0, 0.2)."""
    return np.format_float_positional(np.float32(x), unique=True, trim="0")


# ------------------------------------------------------------------ DiskRollouts (host-side, reference layout)
def _wipe(path):
    """rm -rf `path` as the reference does before every collection -- but the unlink of the previous iteration's shard (0.7 GB
    at 65,536 envs: 50-75 ms of file-system work) happens on a helper thread: the directory is renamed aside first, so the
    caller sees it gone at once."""
    import threading
    aside = "%s.wipe-%d-%d" % (path.rstrip("/"), os.getpid(), threading.get_ident() ^ id(path))
    try:
        os.rename(path, aside)
    except OSError:
        shutil.rmtree(path)
        return
    threading.Thread(target=shutil.rmtree, args=(aside,), kwargs={"ignore_errors": True}, daemon=False).start()


class DiskRollouts:
    """PPO.DiskRollouts(state_data_dir) (src/rollouts_to_disk.jl:23-45): wipes the directory, creates states/,
    starts trajectory.csv with the five-column header."""
    HEADER = ["sample_names", "selected_actions", "selected_action_probabilities", "rewards", "terminal"]
    FINAL = ["sample_names", "selected_actions", "selected_action_probabilities", "returns"]

    def __init__(self, state_data_dir):
        self.state_data_directory = state_data_dir
        self.num_samples = 0
        if os.path.isdir(state_data_dir):                          # prepare_state_data_directory :7-13
            _wipe(state_data_dir)
        os.makedirs(os.path.join(state_data_dir, "states"))
        self.trajectory_filename = os.path.join(state_data_dir, "trajectory.csv")
        with open(self.trajectory_filename, "w", newline="") as f:
            f.write(",".join(self.HEADER) + "\n")
        self._device = None        # BufferRollouts holding the columns when collected on the GPU

    def __len__(self):
        return self.num_samples


def update_(buffer, state, action_probability, action, reward, terminal):
    """PPO.update!(buffer::DiskRollouts, ...) (src/rollouts_to_disk.jl:73-95): one BSON file + one CSV row."""
    from . import PPOError
    if not (0 <= action_probability <= 1):
        raise PPOError(-1, "AssertionError: 0 <= action_probability <= 1")
    if not isinstance(terminal, (bool, np.bool_)):
        raise PPOError(-1, "AssertionError: terminal isa Bool")
    buffer.num_samples += 1
    name = "sample_%d.bson" % buffer.num_samples
    with open(os.path.join(buffer.state_data_directory, "states", name), "wb") as f:
        f.write(bson_encode_state(state))
    with open(buffer.trajectory_filename, "a", newline="") as f:
        f.write("%s,%d,%s,%s,%s\n" % (name, int(action), _fmt32(action_probability), _fmt32(reward),
                                      "true" if terminal else "false"))


def write_returns_to_disk(buffer, discount):
    """src/rollouts_to_disk.jl:106-132: re-read the CSV, compute_returns (on the GPU), rewrite with 4 columns."""
    from . import compute_returns
    with open(buffer.trajectory_filename, newline="") as f:
        rows = list(csv.DictReader(f))
    rewards = np.array([np.float32(r["rewards"]) for r in rows], np.float32)
    terminal = np.array([r["terminal"].strip().lower() == "true" for r in rows], np.uint8)
    ret = compute_returns(rewards, terminal, discount) if len(rows) else np.zeros(0, np.float32)
    with open(buffer.trajectory_filename, "w", newline="") as f:
        f.write(",".join(DiskRollouts.FINAL) + "\n")
        for r, v in zip(rows, ret):
            f.write("%s,%s,%s,%s\n" % (r["sample_names"], r["selected_actions"], r["selected_action_probabilities"],
                                       _fmt32(v)))


def export_reference_layout(disk, device_rollouts, max_samples=100000):
    """Write a finished device rollout in the reference's CSV + per-state BSON layout (small runs only)."""
    from . import PPOError, StateData
    n = len(device_rollouts)
    if n > max_samples:
        raise PPOError(-4, "export_reference_layout: %d samples; one file per state is only meant for small runs" % n)
    st, act = device_rollouts.state_data
    idx = device_rollouts.index()
    st = st.reshape(-1, st.shape[2], st.shape[3])
    act = act.reshape(-1)
    a = device_rollouts.selected_actions.reshape(-1)
    p = device_rollouts.selected_action_probabilities.reshape(-1)
    ret = device_rollouts.rewards.reshape(-1)
    with open(disk.trajectory_filename, "w", newline="") as f:
        f.write(",".join(DiskRollouts.FINAL) + "\n")
        for k, t in enumerate(idx, start=1):
            name = "sample_%d.bson" % k
            with open(os.path.join(disk.state_data_directory, "states", name), "wb") as g:
                g.write(bson_encode_state(StateData(st[t], act[t])))
            f.write("%s,%d,%s,%s\n" % (name, int(a[t]), _fmt32(p[t]), _fmt32(ret[t])))
    disk.num_samples = n


# ------------------------------------------------------------------ DiskDataset
class DiskDataset:
    """src/dataset.jl:1-82.  Reads trajectory.csv (+ BSON states on demand); 1-based indices."""

    def __init__(self, root_directory, trajectory_filename="trajectory.csv", states_dirname="states"):
        from . import PPOError
        self.root_directory = root_directory
        path = os.path.join(root_directory, trajectory_filename)
        if not os.path.isfile(path):
            raise PPOError(-1, "AssertionError: isfile(trajectory_filepath)")            # :7
        with open(path, newline="") as f:
            self.trajectory_df = list(csv.DictReader(f))
        self.states_directory = os.path.join(root_directory, states_dirname)
        if not os.path.isdir(self.states_directory):
            raise PPOError(-1, "AssertionError: isdir(states_directory)")                # :16

    def __len__(self):
        return len(self.trajectory_df)

    def load_sample(self, idx):
        from . import PPOError, StateData
        if not isinstance(idx, (int, np.integer)) or not (1 <= idx <= len(self)):
            raise PPOError(-1, "AssertionError: 1 <= idx <= size(trajectory_df, 1)")     # :32-33
        row = self.trajectory_df[idx - 1]
        path = os.path.join(self.states_directory, row["sample_names"])
        if not os.path.isfile(path):
            raise PPOError(-1, "AssertionError: isfile(state_filepath)")                 # :37
        st = bson_decode_state(open(path, "rb").read())
        if isinstance(st, dict) and "vertex_score" in st:
            m = st["action_mask"].reshape(-1)
            bits = 0
            for q in range(len(m) // 16):
                bits |= (0 if np.isneginf(m[16 * q]) else 1) << q
            st = StateData(st["vertex_score"].astype(np.int8), np.uint32(bits))
        return {"state": st, "selected_action": int(row["selected_actions"]),
                "selected_action_probability": float(np.float32(row["selected_action_probabilities"])),
                "returns": float(np.float32(row["returns"]))}

    def __getitem__(self, idx):
        from . import PPOError, batch_state
        if isinstance(idx, (int, np.integer)):
            return self.load_sample(int(idx))
        if isinstance(idx, (list, tuple, np.ndarray)):
            samples = [self.load_sample(int(i)) for i in idx]                            # :54-72
            return {"state": batch_state([s["state"] for s in samples]),
                    "selected_action": np.array([s["selected_action"] for s in samples], np.int64),
                    "selected_action_probability": np.array([s["selected_action_probability"] for s in samples], np.float32),
                    "returns": np.array([s["returns"] for s in samples], np.float32)}
        raise PPOError(-1, "Dataset index should be Int or Array, got %s" % type(idx).__name__)

    def to_device(self, env):
        """Upload the whole dataset into a device rollout buffer (one column of `len` transitions per env slot is
        not needed: the samples are laid out as T = ceil(len/N) steps of the env's N columns; the tail is padded
        by repeating the last sample and excluded from the dataset index by the caller)."""
        from . import BufferRollouts, PPOError
        n = len(self)
        if n == 0:
            raise PPOError(-1, "AssertionError: empty dataset")
        batch = self[list(range(1, n + 1))]
        N = env.N
        T = -(-n // N)
        pad = T * N - n

        def padded(a):
            return np.concatenate([a, np.repeat(a[-1:], pad, axis=0)]) if pad else a
        vs = padded(batch["state"].vertex_score).reshape(T, N, env.H, env.F)
        am = padded(np.asarray(batch["state"].action_mask, np.uint32)).reshape(T, N)
        ro = BufferRollouts()
        ro.set_columns(env, vs, am, padded(batch["selected_action"]).reshape(T, N),
                       padded(batch["selected_action_probability"]).reshape(T, N),
                       padded(batch["returns"]).reshape(T, N), None)
        return ro, n
