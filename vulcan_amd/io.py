"""Files on either side of the hot path: the mesh the extractor
produces and the RGB-D frames a sequence consists of. Host code; no device work.

write_ply     vulcan::Exporter::Export (src/exporter.cpp:19-71), byte for byte: ASCII PLY,
              `element vertex` with x y z and the red/green/blue debug colouring upstream
              marks "REMOVE" (a grey ramp over z from a fixed 0.35 to the largest z),
              `element face` as `3 i j k`; numbers as a C++ ostream prints floats.
read_ply      the inverse, for tests.
load_depth / save_depth, load_color / save_color
              Image::Load / Save and ColorImage::Load / Save (include/vulcan/image.h:100-133,
              228-253; src/image.cu:213-221,264-273) for the formats that need no OpenCV:
              binary PGM (8 or 16 bit, the depth maps a sensor delivers in millimetres) and binary
              PPM. Load: pixel * scale as float32; Save: pixel * alpha + beta, rounded to
              nearest even and saturated to the integer type, as cv::Mat::convertTo does.
SequenceWriter / SequenceReader
              a directory of depth_%06d.pgm / color_%06d.ppm plus sequence.txt (size,
              intrinsics, depth scale, one depth_to_world matrix + inverse per frame): what stands in for the
              HAL camera the upstream app reads from (apps/vulcan/vulcan.cu:181-232).
"""
import os

import numpy as np


def _g(x):
    """`std::ostream << float`: %g with 6 significant digits."""
    return "%g" % float(np.float32(x))


def write_ply(path, points, faces):
    points = np.asarray(points, dtype=np.float32).reshape(-1, 3)
    faces = np.asarray(faces, dtype=np.int32).reshape(-1, 3)
    out = ["ply", "format ascii 1.0", f"element vertex {len(points)}", "property float x", "property float y",
           "property float z", "property uchar red", "property uchar green", "property uchar blue",
           f"element face {len(faces)}", "property list uchar int vertex_indices", "end_header"]
    # exporter.cpp:38-56: dmax = largest z; dmin is overwritten with 0.35f
    F = np.float32
    dmin = F(0.35)
    dmax = points[:, 2].max() if len(points) else F(0)
    with np.errstate(divide="ignore", invalid="ignore"):
        for p in points:
            ratio = F(F(p[2] - dmin) / F(dmax - dmin))                 # exporter.cpp:62
            value = F(255) * (ratio if ratio < F(1.0) else F(1.0))     # min(1.0f, ratio), math.h:9-15
            color = int(value) if np.isfinite(value) else 0             # int(float) truncates; non-finite is undefined upstream
            out.append(f"{_g(p[0])} {_g(p[1])} {_g(p[2])} {color} {color} {color}")
    for f in faces:
        out.append(f"3 {int(f[0])} {int(f[1])} {int(f[2])}")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")


def read_ply(path):
    with open(path) as fh:
        lines = fh.read().split("\n")
    assert lines[0] == "ply" and lines[1] == "format ascii 1.0"
    nv = int(lines[2].split()[-1])
    end = lines.index("end_header")
    nf = int([l for l in lines[:end] if l.startswith("element face")][0].split()[-1])
    body = lines[end + 1:]
    verts = np.array([[float(t) for t in body[i].split()[:3]] for i in range(nv)], dtype=np.float32).reshape(-1, 3)
    colors = np.array([[int(t) for t in body[i].split()[3:6]] for i in range(nv)], dtype=np.int32).reshape(-1, 3)
    faces = np.array([[int(t) for t in body[nv + i].split()[1:4]] for i in range(nf)], dtype=np.int32).reshape(-1, 3)
    return verts, colors, faces


# ---- Netpbm -------------------------------------------------------------------------

def _read_pnm(path):
    with open(path, "rb") as fh:
        data = fh.read()
    tokens, pos = [], 0
    while len(tokens) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tokens.append(data[pos:end])
        pos = end
    pos += 1                                    # the single whitespace after maxval
    magic, w, h, maxval = tokens[0], int(tokens[1]), int(tokens[2]), int(tokens[3])
    channels = {b"P5": 1, b"P6": 3}[magic]
    dtype = np.dtype(">u2") if maxval > 255 else np.uint8
    pixels = np.frombuffer(data, dtype=dtype, count=w * h * channels, offset=pos).reshape(h, w, channels)
    return pixels.astype(np.uint16 if maxval > 255 else np.uint8)


def _write_pnm(path, pixels):
    pixels = np.asarray(pixels)
    h, w = pixels.shape[:2]
    channels = 1 if pixels.ndim == 2 else pixels.shape[2]
    maxval = 65535 if pixels.dtype == np.uint16 else 255
    with open(path, "wb") as fh:
        fh.write(b"%s\n%d %d\n%d\n" % (b"P5" if channels == 1 else b"P6", w, h, maxval))
        fh.write(pixels.astype(">u2" if maxval > 255 else np.uint8).tobytes())


def _convert_to(values, integer_type, alpha, beta):
    """cv::Mat::convertTo(type, alpha, beta): saturate_cast<T>(cvRound(v * alpha + beta))."""
    v = np.asarray(values, dtype=np.float64) * alpha + beta
    info = np.iinfo(integer_type)
    return np.clip(np.rint(v), info.min, info.max).astype(integer_type)


def load_depth(path, scale=1.0):
    """Image::Load(file, scale) (image.h:100-111): grey PGM -> float32 [h, w]; an RGB file is
    reduced to grey first as cv::cvtColor(RGB2GRAY) does (0.299 R + 0.587 G + 0.114 B)."""
    px = _read_pnm(path)
    if px.shape[2] == 3:
        grey = 0.299 * px[..., 0].astype(np.float64) + 0.587 * px[..., 1] + 0.114 * px[..., 2]
        px = np.rint(grey).astype(px.dtype)[..., None]
    return (px[..., 0].astype(np.float32) * np.float32(scale)).astype(np.float32)


def save_depth(path, image, bits=8, alpha=1.0, beta=0.0):
    """Image::Save(file, CV_8UC1 | CV_16UC1, alpha, beta) (image.cu:213-221)."""
    _write_pnm(path, _convert_to(image, np.uint16 if bits == 16 else np.uint8, alpha, beta))


def load_color(path, scale=1.0):
    """ColorImage::Load(file, scale) (image.h:228-240): PPM -> float32 [h, w, 3] in R, G, B order;
    a grey file is replicated to three channels (cvtColor GRAY2RGB)."""
    px = _read_pnm(path)
    if px.shape[2] == 1:
        px = np.repeat(px, 3, axis=2)
    return (px.astype(np.float32) * np.float32(scale)).astype(np.float32)


def save_color(path, image, bits=8, alpha=1.0, beta=0.0):
    """ColorImage::Save (image.cu:264-273)."""
    _write_pnm(path, _convert_to(image, np.uint16 if bits == 16 else np.uint8, alpha, beta))


# ---- sequences ----------------------------------------------------------------------

class SequenceWriter:
    """depth_%06d.pgm (16 bit, depth / depth_scale), color_%06d.ppm (8 bit, colour * 255) and
    sequence.txt."""

    def __init__(self, directory, width, height, depth_projection, color_projection=None, depth_scale=0.001):
        os.makedirs(directory, exist_ok=True)
        self.directory, self.depth_scale, self.count = directory, float(depth_scale), 0
        k, c = depth_projection, color_projection or depth_projection
        self.header = [f"vulcan-sequence 1", f"size {width} {height}", f"depth_scale {self.depth_scale:.9g}",
                       "depth_projection " + " ".join(f"{float(v):.9g}" for v in (k.fx, k.fy, k.cx, k.cy)),
                       "color_projection " + " ".join(f"{float(v):.9g}" for v in (c.fx, c.fy, c.cx, c.cy))]
        self.poses = []

    def append(self, depth, color=None, depth_to_world=None):
        i = self.count
        save_depth(os.path.join(self.directory, "depth_%06d.pgm" % i), depth, bits=16, alpha=1.0 / self.depth_scale)
        if color is not None:
            save_color(os.path.join(self.directory, "color_%06d.ppm" % i), color, alpha=255.0)
        # matrix and cached inverse, both row-major: a Transform carries its inverse and never
        # recomputes it (transform.h:168-170), so the file keeps both
        if depth_to_world is None:
            m = inv = np.eye(4, dtype=np.float32)
        else:
            m = np.asarray(depth_to_world.matrix(), dtype=np.float32)
            inv = np.asarray(depth_to_world.inverse_matrix(), dtype=np.float32)
        self.poses.append("pose %d " % i + " ".join(f"{float(v):.9g}" for v in np.concatenate([m.reshape(-1), inv.reshape(-1)])))
        self.count += 1

    def close(self):
        with open(os.path.join(self.directory, "sequence.txt"), "w") as fh:
            fh.write("\n".join(self.header + [f"frames {self.count}"] + self.poses) + "\n")


class SequenceReader:
    def __init__(self, directory):
        self.directory = directory
        self.poses, self.inverse_poses = {}, {}
        for line in open(os.path.join(directory, "sequence.txt")):
            t = line.split()
            if not t:
                continue
            if t[0] == "size":
                self.width, self.height = int(t[1]), int(t[2])
            elif t[0] == "depth_scale":
                self.depth_scale = float(t[1])
            elif t[0] == "depth_projection":
                self.depth_projection = tuple(float(v) for v in t[1:5])
            elif t[0] == "color_projection":
                self.color_projection = tuple(float(v) for v in t[1:5])
            elif t[0] == "frames":
                self.count = int(t[1])
            elif t[0] == "pose":
                values = np.array([float(v) for v in t[2:34]], dtype=np.float32)
                self.poses[int(t[1])] = values[:16].reshape(4, 4)
                self.inverse_poses[int(t[1])] = values[16:32].reshape(4, 4)

    def __len__(self):
        return self.count

    def frame(self, i):
        """(depth [h, w] in metres, colour [h, w, 3] in [0, 1] or None, depth_to_world 4x4)."""
        depth = load_depth(os.path.join(self.directory, "depth_%06d.pgm" % i), self.depth_scale)
        path = os.path.join(self.directory, "color_%06d.ppm" % i)
        color = load_color(path, 1.0 / 255.0) if os.path.exists(path) else None
        return depth, color, self.poses.get(i, np.eye(4, dtype=np.float32))
