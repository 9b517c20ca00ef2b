"""Scalar logging of the training loop (the reference logs through xax to TensorBoard, train.py:1783-1790; tensorboard itself is
not installed here). Two sinks, both plain files:
  * `scalars.csv`  - one row per logged step, one column per tag;
  * `events.out.tfevents.<time>.kbj` - a TensorBoard event file written by hand: TFRecord framing (length, masked crc32c of the
    length, payload, masked crc32c of the payload) around protobuf-encoded `Event{wall_time, step, summary{value{tag, simple_value}}}`
    messages, so `tensorboard --logdir` shows the same curves the reference's logger would.
"""
from __future__ import annotations

import csv
import os
import struct
import time
from typing import Dict, Optional

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num: int, wire: int, payload: bytes) -> bytes:
    return _varint((num << 3) | wire) + payload


def _ld(num: int, data: bytes) -> bytes:           # length-delimited field
    return _field(num, 2, _varint(len(data)) + data)


def encode_scalar_event(wall_time: float, step: int, scalars: Dict[str, float]) -> bytes:
    summary = b"".join(_ld(1, _ld(1, tag.encode()) + _field(2, 5, struct.pack("<f", float(v)))) for tag, v in scalars.items())
    return _field(1, 1, struct.pack("<d", wall_time)) + _field(2, 0, _varint(step)) + _ld(5, summary)


def tfrecord(payload: bytes) -> bytes:
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", _masked_crc(head)) + payload + struct.pack("<I", _masked_crc(payload))


class ScalarLogger:
    def __init__(self, log_dir: str, tensorboard: bool = True):
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self._csv_path = os.path.join(log_dir, "scalars.csv")
        self._rows = []
        self._tags = []
        self._tb = None
        if tensorboard:
            self._tb = open(os.path.join(log_dir, f"events.out.tfevents.{int(time.time())}.kbj"), "wb")
            # file_version record first, as TensorBoard's own writer does
            self._tb.write(tfrecord(_field(1, 1, struct.pack("<d", time.time())) + _ld(3, b"brain.Event:2")))
            self._tb.flush()

    def log(self, step: int, scalars: Dict[str, float], wall_time: Optional[float] = None):
        wall_time = time.time() if wall_time is None else wall_time
        for t in scalars:
            if t not in self._tags:
                self._tags.append(t)
        self._rows.append((step, wall_time, dict(scalars)))
        with open(self._csv_path, "w", newline="") as f:     # rewritten whole: tags may appear later (validation scalars)
            w = csv.writer(f)
            w.writerow(["step", "wall_time"] + self._tags)
            for s, t, d in self._rows:
                w.writerow([s, f"{t:.3f}"] + [("" if k not in d else repr(float(d[k]))) for k in self._tags])
        if self._tb:
            self._tb.write(tfrecord(encode_scalar_event(wall_time, step, scalars)))
            self._tb.flush()

    def close(self):
        if self._tb:
            self._tb.close()
            self._tb = None


def read_event_file(path: str):
    """Parse an event file written above back into [(step, {tag: value})] (used by the tests; checks both CRCs)."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0

    def rd_varint(buf, p):
        n = shift = 0
        while True:
            b = buf[p]; p += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, p

    def fields(buf):
        p = 0
        while p < len(buf):
            key, p = rd_varint(buf, p)
            num, wire = key >> 3, key & 7
            if wire == 0:
                v, p = rd_varint(buf, p)
            elif wire == 1:
                v, p = buf[p:p + 8], p + 8
            elif wire == 5:
                v, p = buf[p:p + 4], p + 4
            else:
                n, p = rd_varint(buf, p)
                v, p = buf[p:p + n], p + n
            yield num, wire, v

    while pos < len(data):
        head = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        assert struct.unpack("<I", data[pos + 8:pos + 12])[0] == _masked_crc(head), "length crc"
        payload = data[pos + 12:pos + 12 + n]
        assert struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])[0] == _masked_crc(payload), "payload crc"
        pos += 16 + n
        step, scal = 0, {}
        for num, wire, v in fields(payload):
            if num == 2:
                step = v
            elif num == 5:
                for n2, _, val in fields(v):
                    if n2 == 1:
                        tag, sv = None, None
                        for n3, _, x in fields(val):
                            if n3 == 1:
                                tag = x.decode()
                            elif n3 == 2:
                                sv = struct.unpack("<f", x)[0]
                        scal[tag] = sv
        if scal:
            out.append((step, scal))
    return out
