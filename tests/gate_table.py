#!/usr/bin/env python3
"""The north star's two gates as STATISTICS (VERDICT r3 "next" item 2): for every precision mode of a model,

    embedding gate   max |e(mode) - e(f32)| / max |e(f32)| < 1e-4 over a planted-speaker set, and
    EER gate         |EER(mode) - EER(f32)| < 0.01 % absolute on the same trials,

on >= 3 weight seeds x >= 3 trial lists (50 000 / 37 720 / 50 000 trials with 50 % targets: one flipped trial moves an error rate
by 0.004 - 0.005 %, so every list resolves the gate - a 20 000-trial list, where ONE trial is exactly 0.01 %, cannot: the first
version of the ResNet table showed one such draw) plus one list of 500 000 trials per weight seed.  A mode "passes the EER gate" only if it
passes on EVERY draw.  The exact-f32 extraction is the reference-equivalent one (pinned to the reference's own outputs by the
golden fixtures and to the numpy oracle on utterances of these very sets by tests/test_gpu_eer_gate.py, c4_standin.py,
c5_standin.py).  Scoring chains: cosine (sub-mean, length-norm, dot products: score/process.sh:177-203, score/score.sh:82-97)
for the x-vector and ECAPA (configs[1], configs[2] / [3]); PLDA trained on each mode's own embeddings, LLR
(score/pyplda/plda_base.py:93-136, 248-300) for the ResNet (configs[4]).

    python tests/gate_table.py [--models xvector,ecapa,resnet] [--weight-seeds 0,1,2] [--out gpurun_out/eer_gate_table.json]

Used by bench.py (`parity_grade`: the fastest gate-passing mode per model, in the driver's line), by tests/test_gpu_eer_gate.py,
and stand-alone (the committed table: profiles/r4_eer_gate_table.json).  Test infrastructure: lives under tests/."""

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [p for p in (os.path.join(REPO, "asv-subtools_amd", "pytorch"), REPO, os.path.join(REPO, "tests")) if p not in sys.path]

import numpy as np

GATE_REL, GATE_EER = 1e-4, 0.01

RESNET_CREATION = ("ResNetXvector(%d,10,training=False,resnet_params={'use_se':True,'se_ratio':4,'full_pre_activation':False},"
                   "fc2_params={'nonlinearity':'','bn_params':{'momentum':0.5,'affine':False,'track_running_stats':True}})")
MODELS = {
    # name: (blueprint, creation, scoring chain, planted set (speakers, per speaker, t_lo, t_hi, noise), trials per list)
    "xvector": ("xvector.py", "Xvector(%d,10,training=False)", "cosine", (1177, 4, 200, 500, 0.8), 50_000),
    "ecapa": ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(%d,10,training=False)", "cosine", (1177, 4, 200, 500, 0.1), 37_720),
    "resnet": ("resnet_xvector.py", RESNET_CREATION, "plda", (500, 8, 200, 1000, 0.8), 50_000),
}


def fast_trials(labels, n_trials, seed, target_frac=0.5):
    """Vectorised twin of libs.amd.synth.synth_trials: random (enrol, test) pairs, ~target_frac of them same-speaker pairs of two
    different utterances (500 000 trials take the scalar loop half a minute)."""
    r = np.random.RandomState(int(seed))
    labels = np.asarray(labels)
    n = len(labels)
    order = np.argsort(labels, kind="stable")
    sorted_l = labels[order]
    first = np.searchsorted(sorted_l, labels, side="left")               # start of each utterance's speaker group inside `order`
    count = np.searchsorted(sorted_l, labels, side="right") - first
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)                                            # where utterance i sits inside `order`
    a = r.randint(n, size=n_trials)
    want_tgt = r.rand(n_trials) < target_frac
    step = 1 + r.randint(1 << 30, size=n_trials) % np.maximum(count[a] - 1, 1)
    peer = order[first[a] + (pos[a] - first[a] + step) % count[a]]        # another member of a's group (a itself only for singletons)
    other = r.randint(n, size=n_trials)
    for _ in range(16):
        bad = labels[other] == labels[a]
        if not bad.any():
            break
        other[bad] = r.randint(n, size=int(bad.sum()))
    b = np.where(want_tgt, peer, other)
    keep = (b != a) & (want_tgt | (labels[b] != labels[a]))
    a, b = a[keep], b[keep]
    return a.astype(np.int64), b.astype(np.int64), (labels[a] == labels[b]).astype(np.int64)


class Gates(object):
    """One model: planted set resident on the device, engines per (weight seed, precision mode), EER per trial list."""

    def __init__(self, name, feat_dim=80, device=None, model=None, max_frames=130_000, verbose=False):
        import torch
        import helpers
        from libs.amd import synth
        self.name, self.verbose = name, verbose
        blueprint, creation, self.chain, planted, self.n_trials = MODELS[name]
        self.dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.model = model if model is not None else helpers.build_model(blueprint, creation % feat_dim)
        self.shapes = {k: tuple(v.shape) for k, v in self.model.state_dict().items()}
        n_spk, per_spk, t_lo, t_hi, noise = planted
        mats, self.labels = synth.synth_planted_utts(n_spk, per_spk, feat_dim, t_lo, t_hi, noise)
        self.n_utts = len(mats)
        self.frames = int(sum(m.shape[0] for m in mats))
        order = np.argsort([-m.shape[0] for m in mats], kind="stable")
        self.batches, i = [], 0
        while i < len(order):
            j, frames = i, 0
            while j < len(order) and (j == i or frames + mats[order[j]].shape[0] <= max_frames):
                frames += mats[order[j]].shape[0]
                j += 1
            idx = order[i:j]
            offs = np.concatenate([[0], np.cumsum([mats[k].shape[0] for k in idx])]).astype(np.int32)
            self.batches.append((torch.as_tensor(idx, device=self.dev), torch.from_numpy(np.concatenate([mats[k] for k in idx], axis=0)).to(self.dev), offs))
            i = j
        self.planted = dict(speakers=n_spk, per_speaker=per_spk, frames_lo=t_lo, frames_hi=t_hi, noise=noise, utterances=self.n_utts, frames=self.frames)
        if self.chain == "plda":
            self.train_mask = self.labels < int(round(0.5 * n_spk))     # half of the speakers train the PLDA, the others are scored
            self.eval_labels = self.labels[~self.train_mask]
        else:
            self.train_mask, self.eval_labels = None, self.labels
        self._trials = {}

    def trials(self, seed, n):
        key = (int(seed), int(n))
        if key not in self._trials:
            self._trials[key] = fast_trials(self.eval_labels, n, seed)
        return self._trials[key]

    def load_weights(self, seed):
        import torch
        from libs.amd import synth
        sd = synth.synth_state_dict(self.shapes, int(seed))
        self.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        self.model.cuda(self.dev.index)                  # (load_state_dict / .cuda() drop the engines compiled for the previous weights)
        return sd

    def extract(self, precision):
        import torch
        self.model.amd_precision = precision
        eng = self.model._amd_engine()
        out = torch.empty((self.n_utts, eng.embed_dim), dtype=torch.float32, device=self.dev)
        t0 = time.perf_counter()
        for idx, feats, offs in self.batches:
            out[idx] = eng.extract_device(feats, offs)
        torch.cuda.synchronize(self.dev)
        return out, time.perf_counter() - t0

    def scorer(self, emb):
        """-> f(enrol idx, test idx) -> device scores, for the model's scoring chain on these embeddings"""
        from libs.amd import scoring
        if self.chain == "cosine":
            e = scoring.length_normalize(emb, scoring.mean_vector(emb))
            return lambda ei, ti: scoring.score_trials(e, e, ei, ti)
        x = emb.cpu().numpy()
        mean, within, between = scoring.train_plda(x[self.train_mask], self.labels[self.train_mask], num_iters=10)
        plda = scoring.Plda.from_covariances(mean, within, between)
        t = plda.transform_vectors(x[~self.train_mask])
        return lambda ei, ti: plda.llr_trials(t, t, ei, ti)

    def table(self, precisions, weight_seeds=(0, 1, 2), trial_seeds=(41, 42, 43), big_trials=500_000):
        from libs.amd import scoring
        draws = []
        per_mode = {p: {"embedding_max_rel_err": 0.0, "eer_delta_percent": [], "eer_delta_percent_big": [], "max_abs_score_delta": 0.0} for p in precisions}
        eer_ref_all = []
        for ws in weight_seeds:
            self.load_weights(ws)
            ref, _ = self.extract("f32")
            ref_score = self.scorer(ref)
            ref_max = float(ref.abs().max().item())
            lists = [(ts, self.n_trials) for ts in trial_seeds] + ([(trial_seeds[0] + 1000, big_trials)] if big_trials else [])
            ref_eer = {}
            for ts, n in lists:
                ei, ti, tgt = self.trials(ts, n)
                sc = ref_score(ei, ti)
                ref_eer[(ts, n)] = (float(scoring.eer(sc, tgt)[0]), sc)
            eer_ref_all.append(ref_eer[lists[0]][0])
            for p in precisions:
                emb, secs = self.extract(p)
                rel = float((emb - ref).abs().max().item()) / ref_max
                m = per_mode[p]
                m["embedding_max_rel_err"] = max(m["embedding_max_rel_err"], rel)
                score = self.scorer(emb)
                for ts, n in lists:
                    ei, ti, tgt = self.trials(ts, n)
                    sc = score(ei, ti)
                    d = float(scoring.eer(sc, tgt)[0]) - ref_eer[(ts, n)][0]
                    (m["eer_delta_percent_big"] if n == big_trials else m["eer_delta_percent"]).append(round(d, 4))
                    m["max_abs_score_delta"] = max(m["max_abs_score_delta"], float((sc - ref_eer[(ts, n)][1]).abs().max().item()))
                if self.verbose:
                    print("[gates] %s weights %d %-10s rel %.2e  dEER %s | big %s" % (self.name, ws, p, rel, m["eer_delta_percent"][-len(trial_seeds):],
                                                                                     m["eer_delta_percent_big"][-1:]), file=sys.stderr, flush=True)
        out = {"model": self.name, "scoring": self.chain, "planted_set": self.planted, "weight_seeds": list(weight_seeds), "trial_lists": list(trial_seeds),
               "trials_per_list": self.n_trials, "big_list_trials": big_trials, "eer_f32_percent": [round(v, 3) for v in eer_ref_all], "modes": {}}
        for p, m in per_mode.items():
            worst = max(abs(v) for v in m["eer_delta_percent"]) if m["eer_delta_percent"] else None
            worst_big = max(abs(v) for v in m["eer_delta_percent_big"]) if m["eer_delta_percent_big"] else None
            n_pass = sum(1 for v in m["eer_delta_percent"] if abs(v) < GATE_EER)
            out["modes"][p] = {"embedding_max_rel_err": float("%.3g" % m["embedding_max_rel_err"]), "gate_1e-4": bool(m["embedding_max_rel_err"] < GATE_REL),
                               "eer_delta_percent": m["eer_delta_percent"], "eer_delta_percent_500k": m["eer_delta_percent_big"],
                               "draws_passed": "%d of %d" % (n_pass, len(m["eer_delta_percent"])),
                               "eer_gate_every_draw": bool(worst is not None and worst < GATE_EER and (worst_big is None or worst_big < GATE_EER)),
                               "worst_abs_eer_delta_percent": worst, "worst_abs_eer_delta_percent_500k": worst_big,
                               "max_abs_score_delta": float("%.3g" % m["max_abs_score_delta"])}
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="xvector,ecapa,resnet")
    ap.add_argument("--weight-seeds", default="0,1,2")
    ap.add_argument("--precisions", default="f32x,f16,bf16")
    ap.add_argument("--big-trials", type=int, default=500_000)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available(), "gate_table.py needs a ROCm device"
    res = {}
    for name in args.models.split(","):
        t0 = time.perf_counter()
        g = Gates(name, verbose=True)
        res[name] = g.table(args.precisions.split(","), tuple(int(s) for s in args.weight_seeds.split(",")), big_trials=args.big_trials)
        res[name]["seconds"] = round(time.perf_counter() - t0, 1)
        del g
        torch.cuda.empty_cache()
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
