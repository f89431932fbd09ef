"""On-disk formats either side of the hot path — host-side, integer/byte work.

  FNIRSI-1013D dump ("WAV" by name only: a 15 360-byte record, 208-byte header, two channels of 1500
  uint16-LE samples at byte offsets 1000 and 4000):
      read_fnirsi / parse_header / parse_channel_data / import_fnirsi   == main.py:631-679, 1055-1084
      encode_fnirsi                                                     == export_wav, main.py:1136-1191
      (decode_wav.py:43-76 is the same decoder as a script)
  export_data(last_data, path)                                          == main.py:1763-1781 (JSON / CSV)
  read_wav_pcm16(path)     a real RIFF/WAVE PCM16 reader feeding the STFT path (BASELINE config 3);
                           the reference has none.

Decoding is bit-exact with the reference's per-sample Python loops (same operations in the same order on
float64).  The spectrum of an imported record comes from the device through specengine.calculate_fft.
"""
import csv
import json
import os
import struct

import numpy as np

WAV_TOTAL_SIZE = 15360                                   # main.py:29
FNIRSI_SAMPLES = 1500                                    # main.py:670
# FNIRSI scale tables (main.py:32-41): [value, unit, multiplier]
VOLT_LIST = [[5.0, "V", 1], [2.5, "V", 1], [1.0, "V", 1], [500, "mV", 0.001],
             [200, "mV", 0.001], [100, "mV", 0.001], [50, "mV", 0.001]]
TIME_LIST = ([[v, "S", 1] for v in (50, 20, 10, 5, 2, 1)] + [[v, "mS", 0.001] for v in (500, 200, 100, 50, 20, 10, 5, 2, 1)]
             + [[v, "uS", 1e-6] for v in (500, 200, 100, 50, 20, 10, 5, 2, 1)] + [[v, "nS", 1e-9] for v in (500, 200, 100, 50, 20, 10)])


def read_fnirsi(path_or_bytes):
    """(header[208], [ch1_bytes, ch2_bytes]) — main.py:631-644 (short files give short buffers, as f.read does)."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        b = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as fh:
            b = fh.read()
    return b[:208], [b[1000:4000], b[4000:7000]]


def parse_header(header_bytes):
    """([scale_ch1, scale_ch2], time_multiplier) — main.py:646-665 (invalid indices fall back to entry 0)."""
    volt_scale = []
    for x in range(2):
        i = header_bytes[4 + x * 10]
        volt_scale.append(VOLT_LIST[i] if i < len(VOLT_LIST) else VOLT_LIST[0])
    i = header_bytes[22]
    ts = TIME_LIST[i] if i < len(TIME_LIST) else TIME_LIST[0]
    return volt_scale, ts[2]


def parse_channel_data(data_bytes, scale):
    """1500 samples: (lo + 256 hi - 200) * scale[0] / 50.0 — main.py:667-679.  Returns a python list of
    floats like the reference (import_wav stores the list in last_data, main.py:1084)."""
    if len(data_bytes) < 2 * FNIRSI_SAMPLES:
        raise IndexError("index out of range")                       # what data_bytes[i*2] raises in the reference
    raw = np.frombuffer(bytes(data_bytes[:2 * FNIRSI_SAMPLES]), dtype="<u2").astype(np.int64)
    return ((raw - 200) * scale[0] / 50.0).tolist()


def import_fnirsi(path_or_bytes, json_path=None):
    """import_wav without the widgets (main.py:1061-1084, 1106-1114): decode channel 1, build the time axis and
    the spectrum (device), optionally write the reference's side-car JSON.
    Returns {'last_data': {'t','y','f','Y'}, 'voltage_scale', 'time_scale', 'Fs', 'vpp'}."""
    from . import specengine
    header, bufs = read_fnirsi(path_or_bytes)
    volt_scale, tm = parse_header(header)
    time_scale = next((ts for ts in TIME_LIST if ts[2] == tm), TIME_LIST[0])
    ch1 = parse_channel_data(bufs[0], volt_scale[0])
    n = len(ch1)
    t = np.arange(n) * tm
    fs = 1 / tm
    f, Y = specengine.calculate_fft(t, ch1)
    # the reference builds f from Fs = 1/time_multiplier (main.py:1076, 1081), not from t[1]-t[0]
    f = specengine.shifted_freq_axis(n, fs)
    if json_path:
        with open(json_path, "w") as fh:
            json.dump({"time_multiplier": tm, "voltage_scale": volt_scale[0][0], "voltage_unit": volt_scale[0][1],
                       "time_values": t.tolist(), "signal_values": ch1}, fh, indent=2)
    return {"last_data": {"t": t, "y": ch1, "f": f, "Y": Y}, "voltage_scale": volt_scale[0], "time_scale": time_scale,
            "Fs": fs, "vpp": max(ch1) - min(ch1)}


def _encode_samples(values, scale_value):
    """raw = int(value * 50 / scale + 200) clamped to [0, 65535], little endian — main.py:1151-1160."""
    raw = np.asarray(values, dtype=np.float64) * 50.0 / scale_value + 200
    raw = np.clip(np.trunc(raw), 0, 65535).astype("<u2")             # int() truncates toward zero
    return raw.tobytes()


def encode_fnirsi(t, y, imported_voltage_scale=None, imported_time_scale=None):
    """The 15 360-byte record export_wav writes (main.py:1136-1191): channel 1 only, default scale 1.0 and
    header indices 2 / 2 / 0 for generated signals, linear resampling to 1500 points when len(y) != 1500."""
    y = np.asarray(y, dtype=np.float64)
    if imported_voltage_scale is not None:
        scale_value = imported_voltage_scale[0]
        time_scale = imported_time_scale
        vidx = VOLT_LIST.index(list(imported_voltage_scale))
    else:
        scale_value, time_scale, vidx = 1.0, TIME_LIST[0], 2
    if len(y) != FNIRSI_SAMPLES:
        t_orig = np.linspace(t[0], t[-1], len(y))
        t_new = np.linspace(t[0], t[-1], FNIRSI_SAMPLES)
        y = np.interp(t_new, t_orig, y)
    data = _encode_samples(y, scale_value)
    out = bytearray(WAV_TOTAL_SIZE)
    out[4] = vidx
    out[14] = vidx
    out[22] = TIME_LIST.index(list(time_scale))
    out[1000:1000 + len(data)] = data
    return bytes(out)


def export_data(last_data, path):
    """export_data (main.py:1775-1781): .json -> {key: list} with indent=2, anything else -> CSV of zipped rows."""
    if path.endswith(".json"):
        with open(path, "w") as fh:
            json.dump({k: np.asarray(v).tolist() for k, v in last_data.items()}, fh, indent=2)
    else:
        with open(path, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(last_data.keys())
            w.writerows(zip(*last_data.values()))


V15_HEADER_SIZE, V15_DATA_SIZE = 64, 3000                              # versão15.py:29-30


def read_wav_v15(path_or_bytes):
    """The earlier fork's dump layout (versão15.py:604-640): 64-byte header with the sample rate as a little-endian
    uint32 at bytes 24..27, then up to 1500 int16-LE samples; returns (fs, y) with y = int16.astype(float32) / 32768
    — float32, as the reference casts it (:637).  Same error as the reference for files shorter than the header."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        b = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as fh:
            b = fh.read()
    if len(b) < V15_HEADER_SIZE:
        raise ValueError("Arquivo muito pequeno para ser um WAV do Fnirsi 1014D")
    fs_bytes = b[24:28]
    fs = struct.unpack("<I", fs_bytes)[0] if len(fs_bytes) == 4 else 1000
    raw = b[V15_HEADER_SIZE:V15_HEADER_SIZE + V15_DATA_SIZE]
    if len(raw) % 2:
        raw = raw[:-1]
    return fs, np.frombuffer(raw, dtype="<i2").astype(np.float32) / np.float32(32768.0)


def read_wav_pcm16(path_or_bytes):
    """RIFF/WAVE reader for the STFT path: returns (sample_rate, int16 array [n_frames, channels], C-contiguous
    = interleaved, the layout sa_stft_pcm16 takes).  PCM (format tag 1, or WAVE_FORMAT_EXTENSIBLE with the PCM
    sub-format) at 16 bits only; anything else raises ValueError."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        b = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as fh:
            b = fh.read()
    if len(b) < 12 or b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack_from("<I", b, pos + 4)[0]
        body = b[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            if size < 16:
                raise ValueError("truncated fmt chunk")
            tag, ch, rate, _, _, bits = struct.unpack_from("<HHIIHH", body, 0)
            if tag == 0xFFFE and size >= 26:
                tag = struct.unpack_from("<H", body, 24)[0]
            fmt = (tag, ch, rate, bits)
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)                                 # chunks are word aligned
    if fmt is None or data is None:
        raise ValueError("missing fmt or data chunk")
    tag, ch, rate, bits = fmt
    if tag != 1 or bits != 16 or ch < 1:
        raise ValueError(f"only 16-bit PCM is supported (format tag {tag}, {bits} bits, {ch} channels)")
    n = len(data) // (2 * ch)
    return rate, np.frombuffer(data[:n * 2 * ch], dtype="<i2").reshape(n, ch).copy()


def stft_file(path, nfft=8192, hop=2048, window="blackman", **kw):
    """RIFF/WAVE file -> (sample_rate, mag, peak_bin, peak_mag) through the device STFT (fft_amd.stft_pcm16)."""
    from .batch import stft_pcm16
    rate, pcm = read_wav_pcm16(path)
    return (rate,) + tuple(stft_pcm16(pcm, nfft, hop, window, **kw))
