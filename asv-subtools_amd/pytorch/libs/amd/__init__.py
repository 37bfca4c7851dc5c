# -*- coding:utf-8 -*-
"""MI355X (gfx950) engine behind asv-subtools' extract_embedding boundary.

    capi     ctypes binding of libasv_amd.so (include/asv_amd.h)
    ir       layer-program IR + symbolic recorder of a blueprint's extract_embedding body
    engine   compiles the IR into an asv_net_t, batched extraction on device buffers
    shard    length-balanced utterance sharding + RCCL all-gather of embeddings
    scoring  cosine / PLDA / EER on the device
    synth    deterministic synthetic weights / features (tests, bench)
"""
