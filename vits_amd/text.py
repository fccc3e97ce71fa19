"""Phoneme-id front door: the symbol inventory, the symbol -> id mapping and the filelist row format, so that rows of
the reference's training / validation lists can be synthesised directly.

Reference: text/symbols.py:1-45 (inventory: "_" + Mandarin + Japanese + English + punctuation, 519 symbols),
text/__init__.py:6-14 (cleaned_text_to_sequence: one id per space-separated symbol), data_utils.py:51-75 (row format
``speaker|utterance id|phonemes|durations|f0|energy`` with space-separated per-phoneme values, rows longer than 1400
frames skipped, speaker ids from the config's ``data.spk2id``).

The Mandarin and English parts of the inventory are regular and are generated here: 21 initials plus 76 finals (39
plain ones and the erhua forms that differ from them) with tones 1-5, sorted; ARPAbet consonants plus 15 vowels with stress 0-2, sorted.
The order decides the ids a checkpoint was trained with; a test pins the SHA-256 of the whole inventory."""
from __future__ import annotations

import dataclasses
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import torch

_ZH_INITIALS = "b c ch d f g h j k l m n p q r s sh t x z zh".split()
_ZH_FINALS = ("a ai an ang ao e ei en eng er i ia ian iang iao ie ii iii in ing io iong iou o ong ou u ua uai uan uang "
              "uei uen ueng uo v van ve vn").split()
_EN_VOWELS = "AA AE AH AO AW AY EH ER EY IH IY OW OY UH UW".split()
_EN_CONSONANTS = "B CH D DH F G HH JH K L M N NG P R S SH T TH V W Y Z ZH".split()
# Japanese phones carry no regular structure; common ones first, then those only Japanese uses
_JA = ("ts. f. sh. ry. py. h. p. N. a. m. w. ky. n. d. j. cl. ny. z. o. y. t. u. r. pau ch. e. b. k. g. s. i. "
       "gy. my. hy. br by. v. ty. xx. U. I. dy.").split()
_PUNCTUATION = ["!", "?", "…", ",", ".", "sp"]


def _build_symbols() -> List[str]:
    # erhua forms of every final but "er" itself ("e" + "r" coincides with the final "er")
    finals = set(_ZH_FINALS) | {f + "r" for f in _ZH_FINALS if f != "er"}
    zh = sorted(set(_ZH_INITIALS) | {f + str(t) for f in finals for t in range(1, 6)})
    en = sorted(_EN_CONSONANTS + [v + str(s) for v in _EN_VOWELS for s in range(3)])
    return ["_"] + zh + _JA + en + _PUNCTUATION


symbols: List[str] = _build_symbols()
_symbol_to_id: Dict[str, int] = {s: i for i, s in enumerate(symbols)}
MAX_FRAMES = 1400                      # rows longer than this are skipped (reference data_utils.py:53-55)


def cleaned_text_to_sequence(cleaned_text: Sequence[str]) -> List[int]:
    """One id per symbol; an unknown symbol raises KeyError, as in the reference (text/__init__.py:13)."""
    return [_symbol_to_id[s] for s in cleaned_text]


@dataclasses.dataclass
class FilelistRow:
    speaker: str
    utt_id: str
    phonemes: List[int]
    durations: List[int]               # frames per phoneme
    f0: List[float]                    # Hz per phoneme, 0 = unvoiced
    energy: List[float]

    @property
    def frames(self) -> int:
        return sum(self.durations)


def parse_row(line: str) -> FilelistRow:
    """``speaker|id|phonemes|durations|f0|energy`` -> FilelistRow; the four per-phoneme fields must agree in length
    (the reference asserts the same, data_utils.py:88)."""
    parts = line.rstrip("\n").split("|")
    if len(parts) != 6:
        raise ValueError(f"filelist row needs 6 '|'-separated fields, got {len(parts)}")
    spk, utt, ph, du, f0, en = parts
    row = FilelistRow(spk, utt, cleaned_text_to_sequence(ph.split(" ")), [int(v) for v in du.split(" ")],
                      [float(v) for v in f0.strip().split(" ")], [float(v) for v in en.strip().split(" ")])
    n = len(row.phonemes)
    if not (len(row.durations) == len(row.f0) == len(row.energy) == n):
        raise ValueError(f"{spk}/{utt}: {n} phonemes but {len(row.durations)} durations, {len(row.f0)} f0, "
                         f"{len(row.energy)} energy values")
    return row


def read_filelist(path: str, max_frames: int = MAX_FRAMES) -> List[FilelistRow]:
    rows = []
    with open(path, encoding="utf-8") as f:
        for line in f:
            if line.strip():
                r = parse_row(line)
                if r.frames <= max_frames:
                    rows.append(r)
    return rows


def make_batches(rows: Sequence[FilelistRow], max_batch: int = 64) -> List[List[int]]:
    """Row indices sorted by frame count and cut into batches of at most `max_batch` (little padding per batch; the
    composition does not depend on how many GPUs later share the batches)."""
    order = sorted(range(len(rows)), key=lambda i: (rows[i].frames, i))
    return [order[i:i + max_batch] for i in range(0, len(order), max_batch)]


def collate(rows: Sequence[FilelistRow], spk2id: Dict[str, int]) -> Dict[str, torch.Tensor]:
    """Zero-padded batch tensors in the shapes SynthesizerTrn.infer takes (tensor controls: absolute overrides)."""
    B, P = len(rows), max(len(r.phonemes) for r in rows)
    out = dict(phonemes=torch.zeros(B, P, dtype=torch.int64), lengths=torch.zeros(B, dtype=torch.int64),
               sid=torch.zeros(B, dtype=torch.int64), durations=torch.zeros(B, P, dtype=torch.int64),
               f0=torch.zeros(B, P), energy=torch.zeros(B, P))
    for b, r in enumerate(rows):
        n = len(r.phonemes)
        out["phonemes"][b, :n] = torch.tensor(r.phonemes)
        out["durations"][b, :n] = torch.tensor(r.durations)
        out["f0"][b, :n] = torch.tensor(r.f0)
        out["energy"][b, :n] = torch.tensor(r.energy)
        out["lengths"][b] = n
        out["sid"][b] = spk2id[r.speaker]
    return out


def synthesize_rows(net, rows: Sequence[FilelistRow], spk2id: Dict[str, int], max_batch: int = 64,
                    noise_scale: float = 0.667, use_prosody: bool = True, batches: Optional[Iterable[List[int]]] = None,
                    ) -> Iterator[Tuple[int, torch.Tensor]]:
    """Synthesise filelist rows in length-sorted batches (through SynthesizerTrn.infer_batches, so the front end of
    one batch overlaps the generator of the previous one); yields (row index, waveform [samples]) with the padded tail
    cut off.  use_prosody: pass the rows' durations, F0 and energy as overrides (re-synthesis of the recorded
    prosody); otherwise only the phonemes and the speaker are used and the model predicts the rest."""
    hop = net.total_upsample
    todo = list(batches if batches is not None else make_batches(rows, max_batch))
    group = 16                                   # batches handed to the throughput mode at a time (inputs are uploaded up front)
    for g0 in range(0, len(todo), group):
        part = todo[g0:g0 + group]

        def kwargs():
            for idx in part:
                b = collate([rows[i] for i in idx], spk2id)
                kw = dict(phonemes=b["phonemes"], phonemes_lengths=b["lengths"], sid=b["sid"], noise_scale=noise_scale)
                if use_prosody:
                    kw.update(duration_control=b["durations"], pitch_control=b["f0"], energy_control=b["energy"])
                yield kw
        for idx, (o, x_mask, *_) in zip(part, net.infer_batches(kwargs())):
            n_frames = x_mask.sum(dim=(1, 2)).tolist()
            for j, i in enumerate(idx):
                yield i, o[j, 0, :int(n_frames[j]) * hop]
