"""Bit-stream container of a coded sequence - the MI355X build's mirror of the reference's
``src/utils/stream_helper.py`` (same function names and byte format, so the harness code around
``write_sps`` / ``write_ip`` / ``read_header`` ... drops onto it).

Format (restated from /root/reference/src/utils/stream_helper.py:37-154):
  unit header   1 byte   nal_type << 4 | sps_id            nal_type: 0 SPS, 1 I picture, 2 P picture(s)
  SPS body      varuint height, varuint width
  I/P body      1 byte qp, 1 byte ec_parallel << 1 | reset_feature_memory, varuint length, payload
  varuint       < 2^7: 1 byte 0vvvvvvv;  < 2^14: 2 bytes 10vvvvvv vvvvvvvv;  < 2^30: 4 bytes 11vvvvvv ...
                (big endian; the tag lives in the two top bits of the first byte)
"""
import enum


class NalType(enum.IntEnum):
    NAL_SPS = 0
    NAL_I = 1
    NAL_P = 2


def _put(f, data):
    f.write(data)
    return len(data)


def _get(f, n):
    data = f.read(n)
    if len(data) != n:
        raise EOFError("truncated stream: wanted %d bytes, got %d" % (n, len(data)))
    return data


def write_uint_adaptive(f, value):
    value = int(value)
    if value < 0 or value >= 1 << 30:
        raise ValueError("varuint out of range: %d" % value)
    if value < 1 << 7:
        return _put(f, value.to_bytes(1, "big"))
    if value < 1 << 14:
        return _put(f, (value | 0x8000).to_bytes(2, "big"))
    return _put(f, (value | 0xC0000000).to_bytes(4, "big"))


def read_uint_adaptive(f):
    first = _get(f, 1)[0]
    tag = first >> 6
    if tag < 2:                       # 0vvvvvvv
        return first
    if tag == 2:
        return ((first & 0x3F) << 8) | _get(f, 1)[0]
    rest = _get(f, 3)
    return ((first & 0x3F) << 24) | int.from_bytes(rest, "big")


def write_sps(f, sps):
    if not 0 <= sps["sps_id"] < 16:
        raise ValueError("sps_id must fit 4 bits")
    n = _put(f, bytes([(NalType.NAL_SPS << 4) | sps["sps_id"]]))
    n += write_uint_adaptive(f, sps["height"])
    n += write_uint_adaptive(f, sps["width"])
    return n


def write_ip(f, is_i_frame, sps_id, qp, ec_part, reset_feature_memory, bit_stream):
    if not (0 <= qp < 256 and 0 <= ec_part < 128 and 0 <= sps_id < 16):
        raise ValueError("header field out of range")
    nal = NalType.NAL_I if is_i_frame else NalType.NAL_P
    head = bytes([(nal << 4) | sps_id, qp, (ec_part << 1) | (1 if reset_feature_memory else 0)])
    n = _put(f, head)
    n += write_uint_adaptive(f, len(bit_stream))
    if len(bit_stream):
        n += _put(f, bytes(bit_stream))
    return n


def read_header(f):
    flag = _get(f, 1)[0]
    nal_type = NalType(flag >> 4)
    return {"nal_type": nal_type, "sps_id": flag & 0x0F}


def read_sps_remaining(f, sps_id):
    height = read_uint_adaptive(f)
    width = read_uint_adaptive(f)
    return {"sps_id": sps_id, "height": height, "width": width}


def read_ip_remaining(f):
    qp, flag = _get(f, 2)
    length = read_uint_adaptive(f)
    return qp, flag >> 1, flag & 1, _get(f, length)


class SPSHelper:
    """Allocates / looks up sequence parameter sets by picture size (ids 0..15)."""

    def __init__(self):
        self.spss = []

    def get_sps_by_id(self, sps_id):
        return next((s for s in self.spss if s["sps_id"] == sps_id), None)

    def add_sps_by_id(self, sps):
        for i, s in enumerate(self.spss):
            if s["sps_id"] == sps["sps_id"]:
                self.spss[i] = dict(sps)
                return
        self.spss.append(dict(sps))

    def get_sps_id(self, target):
        """-> (sps_id, is_new)"""
        for s in self.spss:
            if (s["height"], s["width"]) == (target["height"], target["width"]):
                return s["sps_id"], False
        new_id = max((s["sps_id"] for s in self.spss), default=-1) + 1
        if new_id > 15:
            raise ValueError("more than 16 picture sizes in one stream")
        self.spss.append(dict(target, sps_id=new_id))
        return new_id, True
