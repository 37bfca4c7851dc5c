# -*- coding:utf-8 -*-
"""Device side of the pipelined extraction loops of pipeline/onestep/extract_embeddings.py (the stream path AND the sharded path):
buffer sets of {page-locked host input, device input, own engine on its own HIP stream, page-locked result / status word}, one
batch per set in flight, so that reading batch i + 1 and writing batch i - 1 overlap the device work of batch i and consecutive
batches overlap on the device (the small launches at the end of one beside the wide GEMMs at the start of the next).

It is also where the range guard of the default precision mode reaches the scripts (VERDICT r4 / ADVICE r4): 'f32x' splits operands
into IEEE-half halves; an activation beyond +-65504 raises a status bit (asv_net_status).  Engine.extract_batch() checks it; the
scripts go through Engine.extract_device(), which cannot (no host synchronisation) - so every batch submitted here carries an
asynchronous copy of the status word behind its result copy (asv_net_status_async, no extra wait), and a batch that raised the bit
is re-run on the engine's bf16-halves twin ('f32x-bf16': the whole f32 exponent range, ~6e-6 relative) before its vectors are
handed on, with the same RuntimeWarning.  The fp32 reference has no such limit
(/root/reference/pytorch/pipeline/onestep/extract_embeddings.py:70-83 writes whatever the f32 forward gives).
"""

import os
import warnings

import numpy as np

from . import capi


class DeviceSets(object):
    """`n_sets` buffer sets (and `n_engines` engines, each on its own HIP stream) for batches of at most `batch_frames` frames / `batch_utts` utterances.

        sets = DeviceSets(model, batch_frames, batch_utts, dim, max_chunk)
        buf = sets.host_buffer(k)                      # [batch_frames, dim] float32 numpy view of page-locked memory: the reader fills it
        sets.submit(k, offsets, frames)                # async: H2D, extraction, result (+ status) copy; `frames` = rows used of buffer k,
                                                       #        or an ndarray of its own (one utterance longer than the whole buffer)
        sets.input_consumed(k)                         # blocks until buffer k may be refilled (its H2D has finished)
        vectors = sets.finish(k)                       # blocks until the batch is done; range guard; [n, E] float32:
                                                       #   results='host': numpy view of page-locked memory (valid until the next submit(k))
                                                       #   results='device': a CUDA tensor of its own (stays valid)
    ASV_AMD_PIPELINE_ENGINES=1 keeps one engine on one stream (device-resident rate of two: +5 % x-vector, +9 % ECAPA, +19 %
    ResNet34-SE, profiles/r3h_streams.txt)."""

    def __init__(self, model, batch_frames, batch_utts, dim, max_chunk, n_sets=3, results="host", n_engines=2, warm=True):
        import torch
        assert results in ("host", "device")
        self.torch = torch
        self.results = results
        self.max_chunk = int(max_chunk)
        engine = model._amd_engine()
        if dim != engine.feat_dim:
            raise ValueError("the input holds %d-dimensional features, the model expects %d" % (dim, engine.feat_dim))
        self.dev = dev = torch.device("cuda", engine.device_index)
        self.n_sets = n_sets
        # engines / streams are fewer than buffer sets: a third set lets the reader fill batch i + 2 while batch i + 1 waits behind
        # batch i on the device and batch i - 1 is being written (with two sets the read of batch i + 1 only started once batch
        # i - 1 had been written: 173 k instead of 240 k utterances/s through the f32x stream path, profiles/r5a_ark_to_ark.json)
        if os.environ.get("ASV_AMD_PIPELINE_ENGINES", "2") == "1":
            n_engines = 1
        if n_engines <= 1:
            self.engines = [engine]
            self.streams = [torch.cuda.current_stream(dev)]
        else:
            # (cached on the model like the first engine: a second loop over the same model compiles nothing)
            self.engines = [engine] + [model._amd_engine(replica=k) for k in range(1, n_engines)]
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(n_engines)]
        self._next_engine = 0
        self.embed_dim = engine.embed_dim
        self.watch = engine._range_fallback_applies()
        self.host_in = [torch.empty((batch_frames, dim), dtype=torch.float32).pin_memory() for _ in range(n_sets)]
        self.host_np = [t.numpy() for t in self.host_in]
        self.dev_in = [torch.empty((batch_frames, dim), dtype=torch.float32, device=dev) for _ in range(n_sets)]
        self.status_host = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(n_sets)]
        if results == "host":
            self.dev_out = [torch.empty((batch_utts, self.embed_dim), dtype=torch.float32, device=dev) for _ in range(n_sets)]
            self.host_out = [torch.empty((batch_utts, self.embed_dim), dtype=torch.float32).pin_memory() for _ in range(n_sets)]
        self.h2d = [torch.cuda.Event() for _ in range(n_sets)]
        self.done = [torch.cuda.Event() for _ in range(n_sets)]
        self.pending = [None] * n_sets                   # (feats tensor, offsets, out tensor, n, engine index, submission number) of the batch in flight on a set
        self.submitted = 0                               # submit() calls so far
        if warm:
            # one full-size batch of zeros through every engine now: the activation arenas (grown on demand by the first extraction:
            # a few hundred MB of hipMalloc per engine) and the segment tables exist before the first real batch arrives
            utts = max(1, min(batch_utts, batch_frames // 204))
            offs = (np.arange(utts + 1, dtype=np.int64) * (batch_frames // utts)).astype(np.int32)
            with torch.cuda.device(dev):
                self.dev_in[0][:int(offs[-1])].zero_()
                for e, eng in enumerate(self.engines):
                    with torch.cuda.stream(self.streams[e]):
                        eng.extract_device(self.dev_in[0][:int(offs[-1])], offs, max_chunk=self.max_chunk)
                        if self.watch:
                            eng.status_async(self.status_host[0])
                for st in self.streams:
                    st.synchronize()
            self.status_host[0].zero_()
        self.range_reruns = 0
        self.submit_seconds = {}                         # host time inside submit() by piece (ASV_AMD_REPORT_TIMING)

    def host_buffer(self, k):
        return self.host_np[k]

    def submit(self, k, offsets, frames):
        torch = self.torch
        assert self.pending[k] is None, "set %d still holds a batch: finish() it first" % k
        n = len(offsets) - 1
        e = self._next_engine                            # consecutive batches alternate between the engines (each on its own stream: in
        self._next_engine = (e + 1) % len(self.engines)  # order, so the status word copied behind a batch is that batch's alone)
        out = None
        if self.results == "device":
            # a block of the ENGINE stream's pool (the stream that writes it), marked as used by the consumer's stream too: the caching
            # allocator then hands it out again only after both streams have passed its last use (allocated on the caller's
            # stream, a block the collector thread had just freed there could be overwritten under a still-pending cat / index kernel)
            consumer = torch.cuda.current_stream(self.dev)
            with torch.cuda.stream(self.streams[e]):
                out = torch.empty((n, self.embed_dim), dtype=torch.float32, device=self.dev)
            if consumer != self.streams[e]:
                out.record_stream(consumer)
        import time
        t = [time.perf_counter()]
        with torch.cuda.device(self.dev), torch.cuda.stream(self.streams[e]):
            if isinstance(frames, np.ndarray):           # one utterance longer than a whole batch buffer: a pageable copy of its own
                feats = torch.from_numpy(np.ascontiguousarray(frames, dtype=np.float32)).to(self.dev)
            else:
                feats = self.dev_in[k][:frames]
                feats.copy_(self.host_in[k][:frames], non_blocking=True)
            self.h2d[k].record()
            t.append(time.perf_counter())
            if self.results == "host":
                out = self.dev_out[k][:n]
            self.engines[e].extract_device(feats, offsets, max_chunk=self.max_chunk, out=out)
            t.append(time.perf_counter())
            if self.results == "host":
                self.host_out[k][:n].copy_(out, non_blocking=True)
            if self.watch:
                self.engines[e].status_async(self.status_host[k])
            self.done[k].record()
            t.append(time.perf_counter())
        for i, name in enumerate(("h2d", "extract", "d2h")):
            self.submit_seconds[name] = self.submit_seconds.get(name, 0.0) + t[i + 1] - t[i]
        self.submitted += 1
        self.pending[k] = (feats, np.array(offsets, dtype=np.int32), out, n, e, self.submitted)
        return out

    def final_through(self):
        """Every submission numbered <= this (1-based, in submit() order) has been finish()ed - synchronised, range-checked and, where
        flagged, re-run: its result may be read from any stream.  Exact whatever set each batch used (the sharded path releases a
        segment to the gather by it; a rule of "n_sets submissions later" only held while batches rotated over the sets)."""
        live = [p[5] for p in self.pending if p is not None]
        return (min(live) - 1) if live else self.submitted

    def input_consumed(self, k):
        if self.pending[k] is not None:
            self.h2d[k].synchronize()

    def finish(self, k):
        p = self.pending[k]
        if p is None:
            return None
        feats, offsets, out, n, e, _ = p
        self.done[k].synchronize()
        if self.watch and (int(self.status_host[k][0]) & capi.STATUS_HALF_RANGE):
            torch = self.torch
            warnings.warn("asv-subtools_amd: an activation left the IEEE-half range of the f32x mode's operand split (|x| > 65504 or NaN): "
                          "re-running the batch with bf16 operand halves (precision 'f32x-bf16')", RuntimeWarning)
            twin = self.engines[e].wide_range_twin()
            with torch.cuda.device(self.dev), torch.cuda.stream(self.streams[e]):
                twin.extract_device(feats, offsets, max_chunk=self.max_chunk, out=out)
                if self.results == "host":
                    self.host_out[k][:n].copy_(out, non_blocking=True)
                self.streams[e].synchronize()
            self.status_host[k].zero_()
            self.range_reruns += 1
        self.pending[k] = None
        return self.host_out[k][:n].numpy() if self.results == "host" else out

    def flush(self):
        """Finishes every batch in flight (results='device': afterwards every tensor submit() returned is final and safe to read
        from any stream)."""
        for k in range(self.n_sets):
            self.finish(k)
