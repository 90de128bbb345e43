"""utils/streams.py: the streams handed out for overlapping launches are measured to run side by side."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_concurrent_streams_overlap_pairwise():
    from particle_fm_amd.utils.streams import concurrent_streams, overlap
    dev = torch.device("cuda", 0)
    junk = [torch.cuda.Stream(device=dev) for _ in range(5)]  # disturb the runtime's queue bookkeeping first
    del junk[1], junk[2]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # "fewer independent queues than asked for" must not happen for 3 streams
        ss = concurrent_streams(3, dev)
    assert len(ss) == 3 and len({s.cuda_stream for s in ss}) == 3
    for i in range(3):
        for j in range(i + 1, 3):
            assert overlap(ss[i], ss[j], dev)
    assert not overlap(ss[0], ss[0], dev)  # the detector itself: one stream serialises
