"""vits_flow_reverse_noalloc (the entry point for callers that cannot allocate device memory: its work tensors live
in a fixed array in the library's device image) against vits_flow_reverse with a caller workspace: same kernels, so
the same bytes.  Two calls in a row share the array, the second with a smaller input; a size beyond the array is
refused."""
import pytest
import torch

from vits_amd import _native as N
from vits_amd import synthetic

pytestmark = pytest.mark.gpu


def test_flow_reverse_noalloc_matches_workspace_version(synthetic_sd):
    from vits_amd import SynthesizerTrn
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         **cfg.kwargs()).eval()
    net.load_state_dict(synthetic_sd)
    lib, h, dev = N.lib, net._handle, net.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    assert lib.vits_flow_noalloc_max_positions() == 8192
    gen = torch.Generator().manual_seed(5)
    for B, T, lens in ((4, 300, [300, 211, 5, 300]), (2, 61, [61, 40])):            # second call: smaller input
        z0 = (torch.randn(B, cfg.inter_channels, T, generator=gen) * 2.0).to(dev)
        lengths = torch.tensor(lens, dtype=torch.int32, device=dev)
        sid = torch.randint(0, 67, (B,), generator=gen).to(dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        cond = torch.empty(B, lib.vits_cond_rows(h), dtype=torch.float32, device=dev)
        N.check(lib.vits_speaker_cond(h, sid.data_ptr(), B, cond.data_ptr(), err.data_ptr(), stream), "speaker_cond")
        ws = torch.empty(int(lib.vits_flow_workspace_bytes(h, B, T)), dtype=torch.uint8, device=dev)
        z_ref = z0.clone()
        N.check(lib.vits_flow_reverse(h, z_ref.data_ptr(), lengths.data_ptr(), cond.data_ptr(), B, T, ws.data_ptr(),
                                      ws.numel(), stream), "flow_reverse")
        z_new = z0.clone()
        torch.cuda.synchronize(dev)
        before = torch.cuda.memory_allocated(dev)
        free_before = torch.cuda.mem_get_info(dev)[0]             # the driver's view: sees hipMalloc and image loads too
        N.check(lib.vits_flow_reverse_noalloc(h, z_new.data_ptr(), lengths.data_ptr(), cond.data_ptr(), B, T, stream),
                "flow_reverse_noalloc")
        torch.cuda.synchronize(dev)
        assert torch.cuda.memory_allocated(dev) == before
        assert torch.cuda.mem_get_info(dev)[0] >= free_before     # first call included: the array was mapped at load
        assert not torch.equal(z_ref, z0)                                          # the flow did something
        assert torch.equal(z_new, z_ref)                                           # same kernels, same bytes
    # beyond the fixed array: a clear error, nothing launched
    B, T = 5, 2000
    z = torch.zeros(B, cfg.inter_channels, T, device=dev)
    lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
    cond = torch.zeros(B, lib.vits_cond_rows(h), device=dev)
    rc = lib.vits_flow_reverse_noalloc(h, z.data_ptr(), lengths.data_ptr(), cond.data_ptr(), B, T, stream)
    assert rc == -4 and b"8192" in lib.vits_last_error()
    assert lib.vits_flow_reverse_noalloc(h, None, lengths.data_ptr(), cond.data_ptr(), 1, 8, stream) != 0
    assert b"null argument" in lib.vits_last_error()
    torch.cuda.synchronize(dev)
