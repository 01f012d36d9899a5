"""Wire formats of the generated media, as the reference's demo writes them (qwen2.5omni_spider_web.py:122-166):
images through PIL (`image.save(path)`, .jpg / .png), audio as a 16 kHz WAV (`scipy.io.wavfile.write(filename, rate=16000,
data=audio)`, :151-157), video as an 8 fps MP4 (`export_to_video(video, filename, fps=8)`, :131-149). Host I/O only: no GPU
work here. This image has neither cv2 nor imageio (what diffusers' export_to_video drives), so the MP4 is written directly:
an ISO base-media file with one Motion-JPEG video track (sample entry 'jpeg', frames encoded by PIL), which ffmpeg / VLC /
QuickTime decode; `read_mp4_frames` reads such a file back. Nothing is written unless the caller asks (the reference's
absolute /root/autodl-tmp paths are a side effect this build makes optional)."""
from __future__ import annotations

import io
import os
import struct
import tempfile
from datetime import datetime
from typing import List, Optional, Sequence

import numpy as np


def _target(dir_path: Optional[str], ext: str, timestamp: bool = True) -> str:
    """`<dir>/<YYYYmmdd-HHMMSS>/<random><ext>` like save_*_to_local (qwen2.5omni_spider_web.py:122-157)."""
    base = dir_path or tempfile.gettempdir()
    if timestamp:
        base = os.path.join(base, datetime.now().strftime("%Y%m%d-%H%M%S"))
    os.makedirs(base, exist_ok=True)
    return os.path.join(base, next(tempfile._get_candidate_names()) + ext)


def save_image(image, dir_path: Optional[str] = None, ext: str = ".jpg", path: Optional[str] = None) -> str:
    """image: PIL.Image or uint8 HWC array."""
    from PIL import Image
    if not isinstance(image, Image.Image):
        image = Image.fromarray(np.asarray(image, dtype=np.uint8))
    path = path or _target(dir_path, ext)
    image.save(path)
    return path


def save_audio(audio, dir_path: Optional[str] = None, rate: int = 16000, path: Optional[str] = None) -> str:
    """1-D (or [n, channels]) waveform -> RIFF/WAVE with the sample type scipy.io.wavfile.write derives from the dtype:
    float32 -> IEEE float (format tag 3, what the AudioLDM decoder returns), int16 -> PCM 16, uint8 -> PCM 8."""
    a = np.asarray(audio)
    if a.dtype == np.float64:
        a = a.astype(np.float32)
    if a.dtype not in (np.float32, np.int16, np.int32, np.uint8):
        raise ValueError(f"unsupported sample dtype {a.dtype}")
    if a.ndim not in (1, 2):
        raise ValueError("audio must be [samples] or [samples, channels]")
    ch = 1 if a.ndim == 1 else a.shape[1]
    fmt = 3 if a.dtype == np.float32 else 1
    bits = a.dtype.itemsize * 8
    data = np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
    block = ch * bits // 8
    hdr = struct.pack("<HHIIHH", fmt, ch, rate, rate * block, block, bits)
    if fmt != 1:
        hdr += struct.pack("<H", 0)   # cbSize: non-PCM formats use the 18-byte fmt chunk, as scipy writes it
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(hdr)) + hdr
    if fmt == 3:   # non-PCM formats carry a fact chunk (sample frames per channel)
        body += b"fact" + struct.pack("<II", 4, a.shape[0])
    body += b"data" + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b"")
    path = path or _target(dir_path, ".wav")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return path


def _box(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I", 8 + len(payload)) + kind + payload


def _full(kind: bytes, version: int, flags: int, payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def save_video(frames: Sequence, dir_path: Optional[str] = None, fps: int = 8, quality: int = 90, path: Optional[str] = None) -> str:
    """frames: sequence of uint8 HWC arrays / PIL images (one clip, as `.frames[0]` of the text-to-video pipeline)."""
    from PIL import Image
    if len(frames) == 0:
        raise ValueError("no frames")
    jpgs: List[bytes] = []
    size = None
    for fr in frames:
        im = fr if isinstance(fr, Image.Image) else Image.fromarray(np.asarray(fr, dtype=np.uint8))
        im = im.convert("RGB")
        if size is None:
            size = im.size
        elif im.size != size:
            raise ValueError("all frames must have the same size")
        buf = io.BytesIO()
        im.save(buf, format="JPEG", quality=quality)
        jpgs.append(buf.getvalue())
    w, h = size
    n = len(jpgs)
    ts, delta = fps * 1000, 1000                                                   # media timescale: 1000 ticks per frame
    dur_media, dur_movie = n * delta, n * 1000 // fps                              # movie timescale 1000 (ms)
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isomiso2mp41")
    mdat = _box(b"mdat", b"".join(jpgs))
    chunk_off = len(ftyp) + 8
    ident = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)     # unity matrix
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, 1000, dur_movie) + struct.pack(">IH", 0x10000, 0x100) + b"\x00" * 10
                 + ident + b"\x00" * 24 + struct.pack(">I", 2))
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, dur_movie) + b"\x00" * 8 + struct.pack(">HHHH", 0, 0, 0, 0)
                 + ident + struct.pack(">II", w << 16, h << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, ts, dur_media, 0x55C4, 0))
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I", 0) + b"vide" + b"\x00" * 12 + b"VideoHandler\x00")
    entry = (b"\x00" * 6 + struct.pack(">H", 1) + b"\x00" * 16 + struct.pack(">HH", w, h) + struct.pack(">II", 0x480000, 0x480000)
             + struct.pack(">IH", 0, 1) + bytes([10]) + b"Photo-JPEG".ljust(31, b"\x00") + struct.pack(">Hh", 24, -1))
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + _box(b"jpeg", entry))
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1))
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", len(j)) for j in jpgs))
    stco = _full(b"stco", 0, 0, struct.pack(">II", 1, chunk_off))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
    minf = _box(b"minf", _full(b"vmhd", 0, 1, b"\x00" * 8) + dinf + stbl)
    moov = _box(b"moov", mvhd + _box(b"trak", tkhd + _box(b"mdia", mdhd + hdlr + minf)))
    path = path or _target(dir_path, ".mp4")
    with open(path, "wb") as f:
        f.write(ftyp + mdat + moov)
    return path


def read_mp4_frames(path: str):
    """Inverse of save_video for the files it writes: -> (frames as uint8 HWC arrays, fps)."""
    from PIL import Image
    buf = open(path, "rb").read()

    def children(lo, hi):
        out = {}
        while lo + 8 <= hi:
            size, kind = struct.unpack(">I4s", buf[lo:lo + 8])
            if size < 8:
                break
            out[kind] = (lo + 8, lo + size)
            lo += size
        return out

    top = children(0, len(buf))
    trak = children(*children(*top[b"moov"])[b"trak"])
    mdia = children(*trak[b"mdia"])
    stbl = children(*children(*mdia[b"minf"])[b"stbl"])
    lo, _ = mdia[b"mdhd"]
    ts = struct.unpack(">I", buf[lo + 12:lo + 16])[0]
    lo, _ = stbl[b"stts"]
    delta = struct.unpack(">I", buf[lo + 12:lo + 16])[0]
    lo, _ = stbl[b"stsz"]
    n = struct.unpack(">I", buf[lo + 8:lo + 12])[0]
    sizes = struct.unpack(f">{n}I", buf[lo + 12:lo + 12 + 4 * n])
    lo, _ = stbl[b"stco"]
    off = struct.unpack(">I", buf[lo + 8:lo + 12])[0]
    frames = []
    for s in sizes:
        frames.append(np.asarray(Image.open(io.BytesIO(buf[off:off + s])).convert("RGB")))
        off += s
    return frames, ts / delta
