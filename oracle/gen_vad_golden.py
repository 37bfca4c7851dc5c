#!/usr/bin/env python3
"""Generate tests/golden/vad.npz from the REFERENCE's own energy VAD: runtime/extractor/torch_asv_extractor.cc compiled in
place into oracle/_ref/libextractor_ref.so (oracle/Makefile.ref; glog / gflags / yaml-cpp replaced by parse-only stand-ins),
called through extractor_ref_wrap.cc.  Build container only.  The fixture holds the log-energy columns (column 0 of the
feature matrix is all ComputeVadEnergy reads), the option sets and the reference's 0/1 decisions.

    make -C oracle -f Makefile.ref && python oracle/gen_vad_golden.py
"""

import ctypes as C
import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, frames, seed, options): VadEnergyOptions fields of torch_asv_extractor.h:21-27
CASES = [
    ("defaults", 731, 1, dict()),
    ("params_h_defaults", 1500, 2, dict(vad_energy_threshold=5.5)),
    ("no_context", 300, 3, dict(vad_frames_context=0, vad_proportion_threshold=0.6)),
    ("absolute_threshold", 400, 4, dict(vad_energy_mean_scale=0.0, vad_energy_threshold=9.0)),
    ("wide_context", 977, 5, dict(vad_frames_context=7, vad_proportion_threshold=0.5)),
    ("one_frame", 1, 6, dict()),
    ("two_frames", 2, 7, dict(vad_frames_context=3)),
    ("exact_proportion", 64, 8, dict(vad_frames_context=2, vad_proportion_threshold=0.4, vad_energy_mean_scale=0.0, vad_energy_threshold=6.0)),
    ("all_quiet", 50, 9, dict(vad_energy_mean_scale=0.0, vad_energy_threshold=100.0)),
    ("long", 6000, 10, dict(vad_energy_mean_scale=0.7, vad_energy_threshold=2.0)),
]
DEFAULTS = dict(vad_energy_threshold=5.0, vad_energy_mean_scale=0.5, vad_frames_context=2, vad_proportion_threshold=0.12)


def energy_column(frames, seed):
    """Log energies with speech-like bursts: a slow gate between a quiet floor and a loud level + frame noise; values are
    rounded to 1/8 so that comparisons against the threshold include exact ties."""
    r = np.random.RandomState(seed)
    t = np.arange(frames)
    gate = (np.sin(t / 37.0 + seed) + 0.4 * np.sin(t / 11.0) > 0.1).astype(np.float32)
    e = 3.0 + 9.0 * gate + 2.5 * r.standard_normal(frames)
    return (np.round(e * 8.0) / 8.0).astype(np.float32)


def main():
    lib = C.CDLL(os.path.join(REPO, "oracle", "_ref", "libextractor_ref.so"))
    lib.extractor_ref_vad_energy.restype = C.c_int
    out = {}
    meta = []
    for name, frames, seed, kw in CASES:
        o = dict(DEFAULTS, **kw)
        e = energy_column(frames, seed)
        feats = np.zeros((frames, 3), dtype=np.float32)          # other columns are ignored by the reference; not constant on purpose
        feats[:, 0] = e
        feats[:, 1:] = np.random.RandomState(seed + 100).standard_normal((frames, 2))
        voiced = np.full(frames, -1.0, dtype=np.float32)
        rc = lib.extractor_ref_vad_energy(feats.ctypes.data_as(C.c_void_p), frames, 3, C.c_float(o["vad_energy_threshold"]),
                                          C.c_float(o["vad_energy_mean_scale"]), int(o["vad_frames_context"]),
                                          C.c_float(o["vad_proportion_threshold"]), voiced.ctypes.data_as(C.c_void_p))
        assert rc == 0, name
        assert set(np.unique(voiced)) <= {0.0, 1.0}
        out[name + "/energy"] = e
        out[name + "/voiced"] = voiced.astype(np.uint8)
        meta.append(dict(name=name, frames=frames, seed=seed, options=o))
        print("%-22s T=%5d voiced=%5d" % (name, frames, int(voiced.sum())))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "vad.npz"), **out)


if __name__ == "__main__":
    main()
