"""Writers for the recorded-IQ containers the reference's file readers accept (test + demo material).

.raw/.iq : interleaved uint8 IQ                        (raw_reader.cpp:155-158 reads it back as (x-127.38)/128)
.sdr/.wav: RIFF/WAVE, 2 channels, PCM16 at 2 048 000 Hz is what openfiledialog.cpp:140-143 writes
.uff     : XML header (xml_descriptor.cpp:98-240), zero padded, then the payload
"""
import struct

import numpy as np


def to_u8(iq, gain=1.0):
    x = np.empty(2 * len(iq), np.float32)
    x[0::2], x[1::2] = iq.real * gain * 128 + 127.38, iq.imag * gain * 128 + 127.38
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def to_int(iq, bits, gain=1.0):
    x = np.empty(2 * len(iq), np.float64)
    x[0::2], x[1::2] = iq.real * gain, iq.imag * gain
    lim = 2 ** (bits - 1)
    return np.clip(np.rint(x * lim), -lim, lim - 1).astype(np.int64)


def pack_int(v, nbytes, big_endian):
    """int64 array -> packed two's complement bytes."""
    u = (v & ((1 << (8 * nbytes)) - 1)).astype(np.uint64)
    cols = [((u >> np.uint64(8 * k)) & np.uint64(0xFF)).astype(np.uint8) for k in range(nbytes)]
    if big_endian:
        cols = cols[::-1]
    return np.stack(cols, axis=1).reshape(-1)


def write_raw(path, iq, gain=1.0):
    to_u8(iq, gain).tofile(path)


def wav_bytes(payload, rate, bits, fmt_tag=1, channels=2, extensible=False, extra_chunks=(), big_endian=False, open_size=False):
    e = ">" if big_endian else "<"
    block = channels * bits // 8
    if extensible:
        guid = struct.pack(e + "H", fmt_tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack(e + "HHIIHHHHI", 0xFFFE, channels, rate, rate * block, block, bits, 22, bits, 3) + guid
    else:
        fmt = struct.pack(e + "HHIIHH", fmt_tag, channels, rate, rate * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack(e + "I", len(fmt)) + fmt
    for cid, data in extra_chunks:
        body += cid + struct.pack(e + "I", len(data)) + data + (b"\0" if len(data) & 1 else b"")
    body += b"data" + struct.pack(e + "I", 0xFFFFFFFF if open_size else len(payload)) + bytes(payload)
    return (b"RIFX" if big_endian else b"RIFF") + struct.pack(e + "I", 0xFFFFFFFF if open_size else len(body)) + body


def write_sdr(path, iq, rate=2048000, gain=1.0):
    """PCM16 stereo WAV, the .sdr flavour."""
    payload = pack_int(to_int(iq, 16, gain), 2, False)
    with open(path, "wb") as fh:
        fh.write(wav_bytes(payload, rate, 16))


def uff_header(rate, bits, container, ordering, order="IQ", n_elements=0, unit="Hz"):
    val = {"Hz": rate, "KHz": rate // 1000, "MHz": rate // 1000000}[unit]
    ch = "".join('<Channel Value="%s"/>' % c for c in order)
    return ('<?xml version="1.0" encoding="UTF-8"?>\n<SDR>\n <Recorder Name="dabstar_amd tools" Version="1"/>\n'
            ' <Device Name="synth" Model="none"/>\n <Time Value="n/a" Unit="UTC"/>\n'
            ' <!-- recorded IQ -->\n <Sample>\n  <Samplerate Unit="%s" Value="%d"/>\n'
            '  <Channels Bits="%d" Container="%s" Ordering="%s" Amount="2">%s</Channels>\n </Sample>\n'
            ' <Datablocks>\n  <Datablock Count="%d" Number="1" Channel="Channel">\n   <Frequency Value="227360" Unit="KHz"/>\n'
            '   <Modulation Value="DAB"/>\n  </Datablock>\n </Datablocks>\n</SDR>\n'
            % (unit, val, bits, container, ordering, ch, n_elements)).encode()


def write_uff(path, payload, rate, bits, container, ordering="LSB", order="IQ", header_bytes=2048):
    nb = {"int8": 1, "uint8": 1, "int16": 2, "int24": 3, "int32": 4, "float32": 4}[container]
    hdr = uff_header(rate, bits, container, ordering, order, len(payload) // nb)
    assert len(hdr) + 500 <= header_bytes
    with open(path, "wb") as fh:
        fh.write(hdr + b"\0" * (header_bytes - len(hdr)) + bytes(payload))
