"""Phoneme front door (vits_amd/text.py): the symbol inventory has the reference's order (SHA-256 of the joined
inventory, computed from reference text/symbols.py when the golden vectors were made), filelist rows in the reference's
format (data_utils.py:51-75) parse, batch and collate as its loader does, and (GPU) rows synthesised through the front
door equal direct infer calls."""
import hashlib

import pytest
import torch

from vits_amd import synthetic, text

import json
import os

from golden_util import GOLDEN

# written by tools/make_goldens.py from the reference's text/symbols.py
SYMBOLS_SHA256 = json.load(open(os.path.join(GOLDEN, "ORACLE_VS_REFERENCE.json")))["symbols_sha256"]

ROWS = [
    "spkA|u1|sp a1 b ai4 …|3 5 2 7 4|0 210.5 0 180.25 0|1.5 20.0 3.25 18.0 0.5",
    "spkB|u2|k. o. n. o. , AH0 SH sp|6 4 3 9 11 5 2 8|300.1 310.0 0 290.5 0 200.0 0 0|10 11 12 13 0.1 9 8 0.2",
    "spkA|u3|zh uang1 !|4 30 6|0 250.0 0|2.0 30.0 1.0",
]


def test_symbol_inventory_matches_reference_order():
    assert len(text.symbols) == 519 == synthetic.N_SYMBOLS and len(set(text.symbols)) == 519
    assert hashlib.sha256("\n".join(text.symbols).encode()).hexdigest() == SYMBOLS_SHA256
    assert text.symbols[0] == "_" and text.symbols[-1] == "sp" and text.symbols[1] == "a1"
    assert text.cleaned_text_to_sequence(["_", "a1", "sp"]) == [0, 1, 518]
    with pytest.raises(KeyError):
        text.cleaned_text_to_sequence(["not-a-phone"])


def test_rows_parse_batch_and_collate(tmp_path):
    rows = [text.parse_row(r) for r in ROWS]
    assert [r.frames for r in rows] == [21, 48, 40] and rows[1].speaker == "spkB" and rows[1].utt_id == "u2"
    assert rows[0].f0 == [0.0, 210.5, 0.0, 180.25, 0.0]
    with pytest.raises(ValueError):
        text.parse_row("spkA|u1|sp a1|3 5 2|0 1|0 1")           # per-phoneme fields disagree
    with pytest.raises(ValueError):
        text.parse_row("spkA|u1|sp a1|3 5")
    long_row = "spkA|u9|a1 a2|1000 401|100 100|1 1"             # 1401 frames: skipped like the reference does
    p = tmp_path / "list.txt"
    p.write_text("\n".join(ROWS + [long_row]) + "\n", encoding="utf-8")
    got = text.read_filelist(str(p))
    assert len(got) == 3
    assert text.make_batches(got, max_batch=2) == [[0, 2], [1]]
    b = text.collate([got[0], got[1]], {"spkA": 3, "spkB": 7})
    assert b["phonemes"].shape == (2, 8) and b["lengths"].tolist() == [5, 8] and b["sid"].tolist() == [3, 7]
    assert b["phonemes"][0, 5:].eq(0).all() and b["durations"][0].tolist() == [3, 5, 2, 7, 4, 0, 0, 0]
    assert b["durations"].dtype == torch.int64 and b["f0"].dtype == torch.float32


@pytest.mark.gpu
def test_rows_through_front_door_equal_direct_infer(synthetic_sd):
    from vits_amd import SynthesizerTrn
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         **cfg.kwargs()).eval()
    net.load_state_dict(synthetic_sd)
    rows = [text.parse_row(r) for r in ROWS]
    spk2id = {"spkA": 3, "spkB": 7}
    torch.manual_seed(5)
    got = dict(text.synthesize_rows(net, rows, spk2id, max_batch=1))
    assert sorted(got) == [0, 1, 2]
    torch.manual_seed(5)                                        # batches of one, same order => same noise draws
    for i in (0, 2, 1):                                         # length-sorted order: 21, 40, 48 frames
        r = rows[i]
        b = text.collate([r], spk2id)
        o = net.infer(b["phonemes"], b["lengths"], sid=b["sid"], noise_scale=0.667, duration_control=b["durations"],
                      pitch_control=b["f0"], energy_control=b["energy"])[0]
        assert got[i].shape == (r.frames * 512,) and torch.equal(got[i], o[0, 0])
    # predicted prosody: runs, finite, one waveform per row
    out = dict(text.synthesize_rows(net, rows, spk2id, use_prosody=False))
    assert all(torch.isfinite(w).all() and w.numel() % 512 == 0 and w.numel() > 0 for w in out.values())
