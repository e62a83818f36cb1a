"""The torch (device) corpus generator must produce the same bytes as the numpy one."""
import numpy as np

import corpus


def test_torch_generator_equals_numpy():
    import torch
    for first, count in ((0, 96), (70000, 64)):
        a = corpus.make_blocks(first, count)
        b = corpus.make_blocks_torch(torch, first, count, "cpu", chunk=16).numpy()
        assert (a == b).all()
    for name in corpus.CLASSES:
        a = corpus.make_blocks(5, 6, only=name)
        b = corpus.make_blocks_torch(torch, 5, 6, "cpu", only=name).numpy()
        assert (a == b).all(), name


def test_mix_and_isolation():
    counts = corpus.class_counts(4096)
    assert sum(counts.values()) == 4096
    for name, pct in zip(corpus.CLASSES, corpus.MIX):
        assert abs(counts[name] / 4096 * 100 - pct) < 3, (name, counts)
    whole = corpus.make_blocks(0, 64)
    assert (corpus.make_blocks(17, 5) == whole[17:22]).all()
