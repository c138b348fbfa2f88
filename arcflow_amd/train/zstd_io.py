"""Zstandard frames for the prompt-embedding cache (the reference's ``ImagePrompt`` items are zstd-compressed pickles,
image_prompts.py:357-383).  Uses the ``zstandard`` module when it is importable and pyarrow's built-in zstd codec otherwise
(this image has pyarrow, not zstandard); both read and write standard zstd frames, so files are interchangeable."""
import io


def _backend():
    try:
        import zstandard
        return 'zstandard', zstandard
    except ImportError:
        pass
    try:
        import pyarrow as pa
        if pa.Codec.is_available('zstd'):
            return 'pyarrow', pa
    except ImportError:
        pass
    return None, None


def available() -> bool:
    return _backend()[0] is not None


def compress(raw: bytes, level: int = 3) -> bytes:
    kind, mod = _backend()
    if kind == 'zstandard':
        return mod.ZstdCompressor(level=level).compress(raw)
    if kind == 'pyarrow':
        return mod.Codec('zstd', compression_level=level).compress(raw, asbytes=True)
    raise RuntimeError('writing .zst needs the zstandard module or a pyarrow build with the zstd codec')


def decompress(blob: bytes) -> bytes:
    """One or more concatenated zstd frames of unknown decompressed size (streaming read)."""
    kind, mod = _backend()
    if kind == 'zstandard':
        with mod.ZstdDecompressor().stream_reader(io.BytesIO(blob)) as r:
            return r.read()
    if kind == 'pyarrow':
        return mod.CompressedInputStream(mod.BufferReader(blob), 'zstd').read()
    raise RuntimeError('reading .zst needs the zstandard module or a pyarrow build with the zstd codec')
