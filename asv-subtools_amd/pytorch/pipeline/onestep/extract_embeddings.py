# -*- coding:utf-8 -*-
"""Offline embedding extractor - same command line as the reference script
(/root/reference/pytorch/pipeline/onestep/extract_embeddings.py:17-45), so
`pipeline/extract_xvectors_for_pytorch.sh` can call it unchanged:

    extract_embeddings.py [--nnet-config CFG | --model-blueprint PY --model-creation CTOR]
                          [--use-gpu true] [--gpu-id N] model-path feats-rspecifier vectors-wspecifier

Differences are internal: features are read in bulk and embedded in packed ragged batches by
libasv_amd.so instead of one `model.extract_embedding()` call, one H2D copy, ~15 launches and one
blocking D2H per utterance (reference lines 73-83).  Output order and bytes per entry are the same
(`key SP \\0B FV \\4 dim data`).  Any error prints a traceback and exits 1 (reference 85-88) - the
calling shell script greps the log for "Error".

Extra options: --batch-frames / --batch-utts bound a batch; --verbose true restores the reference's
per-utterance "Process utterance for key ..." line (a measurable cost at >100k utterances/s).
"""

import argparse
import os
import sys
import traceback

sys.path.insert(0, "subtools/pytorch")
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))

import numpy as np
import torch

import libs.support.kaldi_io as kaldi_io
import libs.support.utils as utils


def get_args(argv=None):
    parser = argparse.ArgumentParser(description="Extract embeddings form a piece of feats.scp or pipeline")
    parser.add_argument("--nnet-config", type=str, default="", help="This config contains model_blueprint and model_creation.")
    parser.add_argument("--model-blueprint", type=str, default=None, help="A *.py which includes the instance of nnet in this training.")
    parser.add_argument("--model-creation", type=str, default=None, help="A command to create the model class, e.g. Xvector(40,2).")
    parser.add_argument("--use-gpu", type=str, default="true", choices=["true", "false"], help="The MI355X path needs a GPU; 'false' is rejected.")
    parser.add_argument("--gpu-id", type=str, default="", help="Specify a fixed gpu, or select gpu automatically.")
    parser.add_argument("--batch-frames", type=int, default=65536, help="Upper bound of frames per packed batch.")
    parser.add_argument("--batch-utts", type=int, default=1024, help="Upper bound of utterances per packed batch.")
    parser.add_argument("--max-chunk", type=int, default=0, help="Override the model's maxChunk (0 = the decorator's value).")
    parser.add_argument("--verbose", type=str, default="false", choices=["true", "false"])
    parser.add_argument("model_path", metavar="model-path", type=str, help="The model used to extract embeddings.")
    parser.add_argument("feats_rspecifier", metavar="feats-rspecifier", type=str, help="")
    parser.add_argument("vectors_wspecifier", metavar="vectors-wspecifier", type=str, help="")
    return parser.parse_args(argv)


def main(argv=None):
    print(" ".join(sys.argv))
    args = get_args(argv)
    try:
        if args.nnet_config != "":
            model_blueprint, model_creation = utils.read_nnet_config(args.nnet_config)
        elif args.model_blueprint is not None and args.model_creation is not None:
            model_blueprint, model_creation = args.model_blueprint, args.model_creation
        else:
            raise ValueError("Expected nnet_config or (model_blueprint, model_creation) to exist.")
        if not utils.to_bool(args.use_gpu):
            raise RuntimeError("asv-subtools_amd extracts on a ROCm device only (--use-gpu=true); there is no CPU path")

        model = utils.create_model_from_py(model_blueprint, model_creation)
        model.load_state_dict(torch.load(args.model_path, map_location="cpu"), strict=False)
        model = utils.select_model_device(model, args.use_gpu, gpu_id=args.gpu_id)
        model.eval()
        max_chunk = args.max_chunk if args.max_chunk > 0 else None
        verbose = utils.to_bool(args.verbose)

        n_done = 0
        with kaldi_io.open_or_fd(args.feats_rspecifier, "rb") as r, kaldi_io.open_or_fd(args.vectors_wspecifier, "wb") as w:
            for keys, feats, offsets in kaldi_io.read_mat_ark_batched(r, max_frames=args.batch_frames, max_utts=args.batch_utts):
                mats = [feats[offsets[i]:offsets[i + 1]] for i in range(len(keys))]
                emb = model.extract_embedding_batch(mats, max_chunk=max_chunk).numpy()
                for key, vec in zip(keys, emb):
                    if verbose:
                        print("Process utterance for key {0}".format(key))
                    kaldi_io.write_vec_flt(w, np.ascontiguousarray(vec), key=key)
                n_done += len(keys)
        print("Extracted {0} embeddings.".format(n_done))
    except BaseException as e:
        if not isinstance(e, KeyboardInterrupt):
            traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
