"""Minimal writer of librealsense-style ROS bag v2.0 recordings — TEST FIXTURE GENERATOR for
pointcloud_stitching_amd/cli/pcs_bag.h. Written from the published container / message layouts (ROS bag
format 2.0; ROS 1 serialisation of sensor_msgs/Image, sensor_msgs/CameraInfo, geometry_msgs/Transform,
std_msgs/Float32; LZ4 frame format), not from any recorder's source. It is NOT evidence that the reader
handles bags recorded by a real librealsense (none is available here).
"""
import struct

import numpy as np

try:
    import xxhash

    def _xxh32(b):
        return xxhash.xxh32(b, seed=0).intdigest()
except Exception:                                   # checksums are not verified by the reader
    def _xxh32(b):
        return 0


def _field(name, value):
    body = name.encode() + b"=" + value
    return struct.pack("<I", len(body)) + body


def _record(fields, data):
    h = b"".join(_field(k, v) for k, v in fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _string(s):
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def _header(seq, t_ns, frame_id=""):
    return struct.pack("<III", seq, t_ns // 10**9, t_ns % 10**9) + _string(frame_id)


def image_msg(seq, t_ns, arr, encoding, step, w):
    h = arr.shape[0]
    data = arr.tobytes()
    assert len(data) == step * h
    return (_header(seq, t_ns) + struct.pack("<II", h, w) + _string(encoding) + b"\0" +
            struct.pack("<I", step) + struct.pack("<I", len(data)) + data)


MODEL_NAMES = {0: "None", 1: "Modified Brown Conrady", 2: "Inverse Brown Conrady", 3: "Ftheta", 4: "Brown Conrady"}


def camera_info_msg(intr):
    K = [intr.fx, 0, intr.ppx, 0, intr.fy, intr.ppy, 0, 0, 1]
    out = _header(0, 0) + struct.pack("<II", intr.height, intr.width) + _string(MODEL_NAMES[intr.model])
    out += struct.pack("<I", 5) + struct.pack("<5d", *[float(c) for c in intr.coeffs])
    out += struct.pack("<9d", *[float(v) for v in K])
    out += struct.pack("<9d", 1, 0, 0, 0, 1, 0, 0, 0, 1)
    out += struct.pack("<12d", *([0.0] * 12))
    out += struct.pack("<II", 0, 0) + struct.pack("<IIIIB", 0, 0, 0, 0, 0)
    return out


def transform_msg(t, q):
    return struct.pack("<3d", *t) + struct.pack("<4d", *q)


def lz4_block(data):
    """Greedy LZ4 block compressor (hash of 4-byte windows). Small and slow; fine for test sizes."""
    n = len(data)
    out = bytearray()
    table = {}
    i = anchor = 0
    end = n - 12                         # last match must start >= 12 bytes before the end
    mv = memoryview(data)

    def emit(lit_from, lit_to, mlen, offset):
        lit = lit_to - lit_from
        tok_l = min(lit, 15)
        tok_m = min(mlen - 4, 15) if mlen else 0
        out.append((tok_l << 4) | tok_m)
        if lit >= 15:
            r = lit - 15
            while r >= 255:
                out.append(255); r -= 255
            out.append(r)
        out.extend(mv[lit_from:lit_to])
        if mlen:
            out.extend(struct.pack("<H", offset))
            if mlen - 4 >= 15:
                r = mlen - 4 - 15
                while r >= 255:
                    out.append(255); r -= 255
                out.append(r)

    while i < end:
        key = bytes(mv[i:i + 4])
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            limit = n - 5                # keep the last 5 bytes literal
            while i + m < limit and data[cand + m] == data[i + m]:
                m += 1
            emit(anchor, i, m, i - cand)
            i += m
            anchor = i
        else:
            i += 1
    emit(anchor, n, 0, 0)
    return bytes(out)


def lz4_frame(data, block=1 << 16):
    flg = (1 << 6) | (1 << 5) | (1 << 2)          # version 1, independent blocks, content checksum
    bd = 4 << 4                                    # 64 KiB max block
    hdr = bytes([flg, bd])
    out = struct.pack("<I", 0x184D2204) + hdr + bytes([(_xxh32(hdr) >> 8) & 0xFF])
    for o in range(0, len(data), block):
        raw = data[o:o + block]
        comp = lz4_block(raw)
        if len(comp) < len(raw):
            out += struct.pack("<I", len(comp)) + comp
        else:
            out += struct.pack("<I", len(raw) | 0x80000000) + raw
    out += struct.pack("<I", 0) + struct.pack("<I", _xxh32(data))
    return out


def write_bag(path, cfg, frames, depth_units=0.001, compression="none", tf_color=None, frames_per_chunk=2,
              color_encoding="rgb8", dt_ns=33_333_333, color_lag_ns=1_000_000):
    """cfg: pcs_stream_config (intrinsics, bpp, stride). frames: list of (depth HxW uint16, colour bytes).
    tf_color: (t[3], q[4] xyzw) of the colour stream -> reference; depth -> reference is the identity."""
    topics = [
        ("/device_0/sensor_0/Depth_0/image/data", "sensor_msgs/Image"),
        ("/device_0/sensor_0/Depth_0/info/camera_info", "sensor_msgs/CameraInfo"),
        ("/device_0/sensor_0/Depth_0/tf/0", "geometry_msgs/Transform"),
        ("/device_0/sensor_0/option/Depth Units/value", "std_msgs/Float32"),
        ("/device_0/sensor_1/Color_0/image/data", "sensor_msgs/Image"),
        ("/device_0/sensor_1/Color_0/info/camera_info", "sensor_msgs/CameraInfo"),
        ("/device_0/sensor_1/Color_0/tf/0", "geometry_msgs/Transform"),
        ("/file_version", "std_msgs/UInt32"),
    ]

    def conn_record(cid):
        topic, typ = topics[cid]
        data = _field("topic", topic.encode()) + _field("type", typ.encode()) + \
            _field("md5sum", b"0" * 32) + _field("message_definition", b"")
        return _record([("op", b"\x07"), ("conn", struct.pack("<I", cid)), ("topic", topic.encode())], data)

    def msg_record(cid, t_ns, body):
        return _record([("op", b"\x02"), ("conn", struct.pack("<I", cid)),
                        ("time", struct.pack("<II", t_ns // 10**9, t_ns % 10**9))], body)

    t0 = 1_600_000_000 * 10**9
    chunks = []
    # chunk 0: connections + static messages
    c0 = b"".join(conn_record(i) for i in range(len(topics)))
    c0 += msg_record(7, t0, struct.pack("<I", 4))
    c0 += msg_record(1, t0, camera_info_msg(cfg.depth))
    c0 += msg_record(5, t0, camera_info_msg(cfg.color))
    c0 += msg_record(2, t0, transform_msg((0, 0, 0), (0, 0, 0, 1)))
    tc, qc = tf_color if tf_color else ((0, 0, 0), (0, 0, 0, 1))
    c0 += msg_record(6, t0, transform_msg(tc, qc))
    if depth_units is not None:
        c0 += msg_record(3, t0, struct.pack("<f", depth_units))
    chunks.append(c0)
    H, W = cfg.depth.height, cfg.depth.width
    cur = b""
    for k, (depth, color) in enumerate(frames):
        t = t0 + (k + 1) * dt_ns
        d = np.ascontiguousarray(depth, np.uint16).reshape(H, W)
        c = np.ascontiguousarray(color, np.uint8).reshape(cfg.color.height, cfg.color_stride)
        cur += msg_record(0, t, image_msg(k, t, d, "mono16", W * 2, W))
        cur += msg_record(4, t + color_lag_ns, image_msg(k, t + color_lag_ns, c, color_encoding, cfg.color_stride, cfg.color.width))
        if (k + 1) % frames_per_chunk == 0:
            chunks.append(cur); cur = b""
    if cur:
        chunks.append(cur)

    with open(path, "wb") as f:
        f.write(b"#ROSBAG V2.0\n")
        # bag header record padded to 4096 bytes; index_pos is filled in afterwards
        def bag_header(index_pos):
            fields = [("op", b"\x03"), ("index_pos", struct.pack("<Q", index_pos)),
                      ("conn_count", struct.pack("<I", len(topics))), ("chunk_count", struct.pack("<I", len(chunks)))]
            h = b"".join(_field(k, v) for k, v in fields)
            pad = 4096 - 4 - len(h) - 4
            return struct.pack("<I", len(h)) + h + struct.pack("<I", pad) + b" " * pad
        f.write(bag_header(0))
        for body in chunks:
            data = lz4_frame(body) if compression == "lz4" else body
            f.write(_record([("op", b"\x05"), ("compression", compression.encode()),
                             ("size", struct.pack("<I", len(body)))], data))
            # one (empty) index record after each chunk, as rosbag writes them; the reader ignores these
            f.write(_record([("op", b"\x04"), ("ver", struct.pack("<I", 1)), ("conn", struct.pack("<I", 0)),
                             ("count", struct.pack("<I", 0))], b""))
        index_pos = f.tell()
        for i in range(len(topics)):
            f.write(conn_record(i))
        f.seek(13)
        f.write(bag_header(index_pos))
