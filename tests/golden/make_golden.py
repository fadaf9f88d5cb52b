"""Generates tests/golden/* from DATA files the reference holds (run in the build container only;
/root/reference does not exist on the GPU box).  No reference source text is copied: only
  - output/trajectory.csv                  (known-answer returns)
  - output/states/sample_1.bson            (183-byte wire-format fixture)
  - test/output/poly-30-policy.bson        (checkpoint wire-format fixture, copied as is)
  - test/output/*.bson                     (trained Float32 policy weights),
    decoded with pymongo's `bson` (a pure data decoder, executes nothing) and re-saved as .npz
    in flat Flux order (W1,b1,W2,b2,W3,b3; W [out,in] column-major).
Known answers that exist only as printed values in the tutorial notebook are written into
known_answers.json by hand with their source line.
"""
import json
import os
import shutil

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def decode_policy(path):
    import bson
    d = bson.decode(open(path, "rb").read())
    arrays = []

    def walk(x):
        if isinstance(x, dict):
            if x.get("tag") == "array" and isinstance(x.get("data"), (bytes, bytearray)):
                name = x["type"]["name"][-1]
                assert name == "Float32", name
                arrays.append((tuple(x["size"]), np.frombuffer(x["data"], "<f4").copy()))
                return
            for k, v in x.items():
                if k != "_backrefs":
                    walk(v)
        elif isinstance(x, list):
            for v in x:
                walk(v)

    walk(d)
    # BSON arrays are column-major with `size` as Julia dims -> already Flux flat order
    flat = np.concatenate([a for (_, a) in arrays]).astype(np.float32)
    shapes = [list(s) for (s, _) in arrays]
    return flat, shapes


def main():
    shutil.copy(os.path.join(REF, "output/trajectory.csv"), os.path.join(OUT, "trajectory.csv"))
    shutil.copy(os.path.join(REF, "output/states/sample_1.bson"), os.path.join(OUT, "sample_1.bson"))
    # one trained-policy checkpoint as BSON.jl wrote it (data file; wire-format fixture of the checkpoint interop)
    shutil.copy(os.path.join(REF, "test/output/poly-30-policy.bson"), os.path.join(OUT, "poly-30-policy.bson"))
    for name in ["catmull-clark-policy-l4", "poly-30-policy", "catmull-clark-policy"]:
        flat, shapes = decode_policy(os.path.join(REF, "test/output", name + ".bson"))
        sh = np.array([s + [0] * (2 - len(s)) for s in shapes], np.int64)   # vectors -> [n,0]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), params=flat, shapes=sh)
        print(name, flat.size, shapes)
    known = {
        "returns_trajectory_csv": {"source": "output/trajectory.csv:1-7", "rewards": [1, 1, 1, 1, 1, 1],
                                   "terminal": [0, 0, 0, 0, 0, 1], "discount": 1.0,
                                   "returns": [6.0, 5.0, 4.0, 3.0, 2.0, 1.0], "action": 4, "prob": 0.5},
        "returns_test_env": {"source": "test/test_rollout_buffer.jl:4-50", "episodes": 10, "horizon": 10,
                             "reward": 1.0, "discount": 1.0, "returns_per_episode": [10, 9, 8, 7, 6, 5, 4, 3, 2, 1]},
        "index_to_action_triangle": {"source": "examples/triangle/single-flip/learn_flip.ipynb:19669-19699",
                                     "actions_per_edge": 2, "edges": 3,
                                     "cases": [[5, [1, 3, 1]], [9, [2, 2, 1]]]},
        "mask_pattern": {"source": "examples/triangle/single-flip/learn_flip.ipynb:469-519",
                         "active": [1, 1, 0, 0], "per_quad": 6, "expect_zero": 12, "expect_neginf": 12},
        "philox4x32_10_kat": {"source": "Random123 kat_vectors (Salmon et al. SC'11)",
                              "cases": [
                                  {"ctr": [0, 0, 0, 0], "key": [0, 0],
                                   "out": ["6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8"]},
                                  {"ctr": [4294967295, 4294967295, 4294967295, 4294967295],
                                   "key": [4294967295, 4294967295],
                                   "out": ["408f276d", "41c83b0e", "a20bc7c6", "6d5451fd"]},
                                  {"ctr": [608135816, 2242054355, 320440878, 57701188],
                                   "key": [2752067618, 698298832],
                                   "out": ["d16cfe09", "94fdcceb", "5001e420", "24126ea1"]}]},
    }
    json.dump(known, open(os.path.join(OUT, "known_answers.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
