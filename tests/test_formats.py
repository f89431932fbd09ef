"""CPU: on-disk formats bit-exact against files/arrays produced by the reference's own
methods (tests/golden/ref_formats.*, made by tools/gen_golden_formats.py).  Integer / byte work: exact."""
import json
import os
import struct

import numpy as np
import pytest

from fft_amd import formats as F


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_formats.npz")), json.load(open(os.path.join(golden_dir, "ref_formats.json"), encoding="utf-8"))


def test_fnirsi_decode_bit_exact(gold):
    z, meta = gold
    for name in ("f0", "f1", "f2"):
        file = z[name + "_file"].tobytes()
        header, bufs = F.read_fnirsi(file)
        m = meta["cases"][name]
        assert len(header) == m["header_len"] and [len(b) for b in bufs] == m["buf_lens"]
        vs, tm = F.parse_header(header)
        assert vs == m["volt_scale"] and tm == m["time_multiplier"]
        for ch, b in enumerate(bufs):
            got = F.parse_channel_data(b, vs[ch])
            assert isinstance(got, list) and len(got) == 1500 and isinstance(got[0], float)
            np.testing.assert_array_equal(np.asarray(got), z[f"{name}_ch{ch + 1}"])
    with pytest.raises(IndexError):
        F.parse_channel_data(b"\x00" * 100, F.VOLT_LIST[0])


def test_fnirsi_encode_bit_exact(gold, tmp_path):
    z, meta = gold
    for name in ("e0", "e1", "e3"):
        out = F.encode_fnirsi(z[name + "_t"], z[name + "_y"])
        assert len(out) == F.WAV_TOTAL_SIZE
        np.testing.assert_array_equal(np.frombuffer(out, dtype=np.uint8), z[name + "_file"], err_msg=name)
    m = meta["cases"]["e2"]
    out = F.encode_fnirsi(z["e2_t"], z["e2_y"].tolist(), m["imported_voltage_scale"], m["imported_time_scale"])
    np.testing.assert_array_equal(np.frombuffer(out, dtype=np.uint8), z["e2_file"])
    # decode(encode(x)) returns x quantised to the oscilloscope's 1/50-division grid
    hdr, bufs = F.read_fnirsi(out)
    vs, _ = F.parse_header(hdr)
    back = np.asarray(F.parse_channel_data(bufs[0], vs[0]))
    assert np.abs(back - z["e2_y"]).max() <= vs[0][0] / 50.0


def test_export_data_text_identical(gold, tmp_path):
    z, _ = gold
    last = {k: z["x_" + k] for k in ("t", "y", "f", "Y")}
    for ext in ("json", "csv"):
        p = str(tmp_path / ("out." + ext))
        F.export_data(last, p)
        assert open(p, "rb").read() == z["x_" + ext].tobytes(), ext


def _riff(rate, pcm, extensible=False, extra_chunk=True):
    ch = pcm.shape[1]
    if extensible:
        fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, ch, rate, rate * ch * 2, ch * 2, 16, 22, 16, 3, 1) + b"\x00" * 14
    else:
        fmt = struct.pack("<HHIIHH", 1, ch, rate, rate * ch * 2, ch * 2, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunk:
        body += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"          # odd-sized chunk + pad byte
    data = pcm.astype("<i2").tobytes()
    body += b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_riff_pcm16_reader(tmp_path):
    rng = np.random.default_rng(1)
    for ch in (1, 2, 3):
        pcm = rng.integers(-32768, 32768, size=(1000, ch)).astype(np.int16)
        for ext in (False, True):
            rate, got = F.read_wav_pcm16(_riff(44100, pcm, ext))
            assert rate == 44100 and got.dtype == np.int16 and got.flags.c_contiguous
            np.testing.assert_array_equal(got, pcm)
    p = tmp_path / "a.wav"
    p.write_bytes(_riff(8000, pcm))
    assert F.read_wav_pcm16(str(p))[0] == 8000
    with pytest.raises(ValueError):
        F.read_wav_pcm16(b"RIFX" + b"\x00" * 40)
    bad = bytearray(_riff(8000, pcm, extra_chunk=False))
    bad[20] = 3                                                       # IEEE float tag
    with pytest.raises(ValueError):
        F.read_wav_pcm16(bytes(bad))
    with pytest.raises(ValueError):                                   # the FNIRSI dump is not RIFF
        F.read_wav_pcm16(bytes(F.WAV_TOTAL_SIZE))
