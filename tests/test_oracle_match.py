import numpy as np


def test_match_oracle_first_maximum_and_chunking(synth):
    from oracle import match
    g = synth.make_gallery(5000)
    g[4000] = g[123]  # duplicate row: the FIRST index must win (std::max_element, arcface.cpp:210)
    q = synth.make_queries(g, [123, 4000, 77])
    idx, sim = match.top1(q, g, chunk=1000)
    assert idx.tolist() == [123, 123, 77]
    full = match.similarity(q, g)
    assert full.shape == (3, 5000) and np.array_equal(full.argmax(1), idx)
    assert np.allclose(sim, full.max(1)) and sim.min() > 0.9
