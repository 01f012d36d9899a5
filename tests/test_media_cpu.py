"""Wire formats of saved media (qwen2.5omni_spider_web.py:122-166): WAV @ 16 kHz as scipy.io.wavfile.write lays it out,
8 fps MP4 (Motion-JPEG track) and PIL images. Host I/O, no GPU."""
import struct

import numpy as np
import pytest

from spider_amd import media


def test_wav_float32_matches_scipy_writer_and_reads_back(tmp_path):
    from scipy.io import wavfile
    a = (np.random.default_rng(0).standard_normal(80000) * 0.1).astype(np.float32)   # 5 s @ 16 kHz, the AudioLDM output
    p = media.save_audio(a, path=str(tmp_path / "a.wav"))
    rate, back = wavfile.read(p)
    assert rate == 16000 and back.dtype == np.float32 and np.array_equal(back, a)
    wavfile.write(str(tmp_path / "ref.wav"), 16000, a)
    assert open(p, "rb").read() == open(tmp_path / "ref.wav", "rb").read()             # byte-identical to the reference's writer
    i16 = (a * 32767).astype(np.int16)
    rate, back = wavfile.read(media.save_audio(np.stack([i16, -i16], 1), path=str(tmp_path / "s.wav")))
    assert back.shape == (80000, 2) and np.array_equal(back[:, 0], i16)
    with pytest.raises(ValueError):
        media.save_audio(np.zeros((2, 2, 2), np.float32), path=str(tmp_path / "x.wav"))


def test_mp4_roundtrip_and_box_structure(tmp_path):
    yy, xx = np.mgrid[0:40, 0:72]
    base = np.stack([2 * xx, 3 * yy, xx + yy], -1).astype(np.uint8)               # smooth image: JPEG keeps it within a few levels
    frames = [np.clip(base.astype(np.int32) + 3 * i, 0, 255).astype(np.uint8) for i in range(16)]   # 16 frames like zeroscope
    p = media.save_video(frames, path=str(tmp_path / "v.mp4"), fps=8, quality=95)
    buf = open(p, "rb").read()
    assert buf[4:8] == b"ftyp" and b"moov" in buf and b"jpeg" in buf
    size0 = struct.unpack(">I", buf[:4])[0]
    assert buf[size0 + 4:size0 + 8] == b"mdat"
    back, fps = media.read_mp4_frames(p)
    assert fps == 8 and len(back) == 16 and back[0].shape == (40, 72, 3)
    err = np.abs(back[5].astype(np.int32) - frames[5].astype(np.int32)).mean()
    assert err < 4, err           # lossy, but the right frame in the right order
    assert np.abs(back[15].astype(np.int32).mean() - frames[15].mean()) < 2
    with pytest.raises(ValueError):
        media.save_video([], path=str(tmp_path / "e.mp4"))
    with pytest.raises(ValueError):
        media.save_video([frames[0], frames[0][:20]], path=str(tmp_path / "e.mp4"))


def test_image_and_default_layout(tmp_path):
    from PIL import Image
    img = np.random.default_rng(2).integers(0, 255, (32, 32, 3), dtype=np.uint8)
    p = media.save_image(img, dir_path=str(tmp_path), ext=".png")
    assert p.startswith(str(tmp_path)) and p.endswith(".png") and np.array_equal(np.asarray(Image.open(p)), img)
    assert media.save_image(Image.fromarray(img), dir_path=str(tmp_path)).endswith(".jpg")
