// meshing.cpp — Extractor, Exporter, image files and sequences (ref: src/extractor.cu,
// src/exporter.cpp, include/vulcan/image.h:100-133,228-253, src/image.cu:213-221,264-273).
#include <vulcan/meshing.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include <vulcan/exception.h>
#include <vulcan/image.h>
#include <vulcan/math.h>
#include <vulcan/sequence.h>
#include <vulcan/tsdf_volume.h>

namespace vulcan
{

// ---- Extractor -------------------------------------------------------------------

Extractor::Extractor(std::shared_ptr<const Volume> volume) :
  volume_(volume),
  all_allocated_(false),
  interpolate_(true),
  skipped_(0)
{
  VULCAN_ASSERT_MSG(volume_, "extractor needs a volume");
  counts_.Resize(4);
}

std::shared_ptr<const Volume> Extractor::GetVolume() const { return volume_; }

void Extractor::ResizeMesh(DeviceMesh& mesh) const
{
  // extractor.cu:700-716, for the blocks that can be listed
  const vk_volume v = volume_->ToVk();
  size_t blocks = all_allocated_ ? (size_t)v.main_block_count + v.excess_block_count : volume_->GetVisibleBlocks().GetSize();
  if (blocks == 0) blocks = 1;
  const size_t max_points = 3 * 512 * blocks;
  const size_t max_faces = 5 * 512 * blocks;
  if (mesh.points.GetCapacity() < max_points) mesh.points.Reserve(max_points);
  if (mesh.faces.GetCapacity() < max_faces) mesh.faces.Reserve(max_faces);
  mesh.points.Resize(0);
  mesh.faces.Resize(0);
}

void Extractor::Extract(DeviceMesh& mesh) const
{
  const vk_volume v = volume_->ToVk();
  const size_t bytes = vk_extract_workspace_bytes(v.main_block_count, v.excess_block_count);
  if (workspace_.GetSize() < bytes) workspace_.Resize(bytes);
  ResizeMesh(mesh);
  // capped so that a large table does not ask for tens of GiB up front: a second call with the
  // reported totals follows when the first capacities did not suffice
  size_t point_capacity = min(mesh.points.GetCapacity(), (size_t)1 << 24);
  size_t face_capacity = min(mesh.faces.GetCapacity(), (size_t)1 << 25);
  int counts[4] = {0, 0, 0, 0};
  for (int attempt = 0; attempt < 2; ++attempt)
  {
    if (mesh.points.GetCapacity() < point_capacity) mesh.points.Reserve(point_capacity);
    if (mesh.faces.GetCapacity() < face_capacity) mesh.faces.Reserve(face_capacity);
    VK_ASSERT(vk_extract_mesh(&v, all_allocated_ ? 1 : 0, interpolate_ ? 1 : 0,
        reinterpret_cast<float*>(mesh.points.GetData()), (int32_t)point_capacity,
        reinterpret_cast<int32_t*>(mesh.faces.GetData()), (int32_t)face_capacity, counts_.GetData(),
        workspace_.GetData(), Device::GetStream()));
    counts_.CopyToHost(counts);      // the only readback: four totals
    if ((size_t)counts[0] <= point_capacity && (size_t)counts[1] <= face_capacity) break;
    point_capacity = counts[0];
    face_capacity = counts[1];
  }
  mesh.points.Resize(counts[0]);
  mesh.faces.Resize(counts[1]);
  skipped_ = counts[2];
}

void Extractor::Extract(Mesh& mesh) const
{
  DeviceMesh device;
  Extract(device);
  mesh.points.resize(device.points.GetSize());
  mesh.faces.resize(device.faces.GetSize());
  if (!mesh.points.empty()) device.points.CopyToHost(mesh.points.data());
  if (!mesh.faces.empty()) device.faces.CopyToHost(mesh.faces.data());
}

// ---- Exporter --------------------------------------------------------------------

Exporter::Exporter(const std::string& file) : file_(file) {}

const std::string& Exporter::GetFile() const { return file_; }

// ASCII PLY with the bytes of upstream's writer (ref: exporter.cpp:19-71; tests/test_gpu_extract.py and
// host_tests compare files byte for byte): the header below, then per vertex "x y z g g g" with the
// coordinates in the stream's default float format (= %g) and g a grey value that upstream derives from z as
// a debugging aid — 255 * min(1, (z - 0.35) / (zmax - 0.35)), truncated — then per face "3 i j k".
// The whole file is formatted into one buffer and written once.
namespace
{
const char* const kPlyHeader[] = {
    "property float x", "property float y", "property float z",
    "property uchar red", "property uchar green", "property uchar blue"};

void append_number(std::string& text, float value)
{
  char field[32];
  text.append(field, (size_t)std::snprintf(field, sizeof(field), "%g", value));
}
void append_number(std::string& text, long long value)
{
  char field[32];
  text.append(field, (size_t)std::snprintf(field, sizeof(field), "%lld", value));
}
}  // namespace

void Exporter::Export(const Mesh& mesh) const
{
  const size_t vertices = mesh.points.size(), triangles = mesh.faces.size();
  std::string text;
  text.reserve(160 + 48 * vertices + 40 * triangles);
  text += "ply\nformat ascii 1.0\nelement vertex ";
  append_number(text, (long long)vertices);
  text += '\n';
  for (const char* line : kPlyHeader) { text += line; text += '\n'; }
  text += "element face ";
  append_number(text, (long long)triangles);
  text += "\nproperty list uchar int vertex_indices\nend_header\n";

  // the grey ramp's far end is the largest z of the mesh; its near end is fixed (exporter.cpp:56)
  const float near_z = 0.35f;
  float far_z = 0.0f;
  for (size_t i = 0; i < vertices; ++i)
    if (i == 0 || mesh.points[i][2] > far_z) far_z = mesh.points[i][2];

  for (size_t i = 0; i < vertices; ++i)
  {
    const Vector3f& p = mesh.points[i];
    const float ramp = 255 * min(1.0f, (p[2] - near_z) / (far_z - near_z));
    const long long grey = std::isfinite(ramp) ? (long long)int(ramp) : 0;   // upstream: undefined when far_z == near_z
    for (int axis = 0; axis < 3; ++axis) { append_number(text, p[axis]); text += ' '; }
    append_number(text, grey);  text += ' ';
    append_number(text, grey);  text += ' ';
    append_number(text, grey);  text += '\n';
  }
  for (size_t i = 0; i < triangles; ++i)
  {
    text += '3';
    for (int corner = 0; corner < 3; ++corner) { text += ' '; append_number(text, (long long)mesh.faces[i][corner]); }
    text += '\n';
  }

  std::ofstream out(file_, std::ios::binary);
  VULCAN_ASSERT(out.is_open());
  out.write(text.data(), (std::streamsize)text.size());
}

// ---- Netpbm files ----------------------------------------------------------------

namespace
{

struct Pnm
{
  int width, height, channels, maxval;
  std::vector<unsigned short> pixels;   // row-major, channel-interleaved
};

Pnm ReadPnm(const std::string& file)
{
  std::ifstream in(file, std::ios::binary);
  VULCAN_ASSERT_MSG(in.is_open(), "unable to load file");
  std::string tokens[4];
  for (int i = 0; i < 4;)
  {
    const int c = in.peek();
    VULCAN_ASSERT_MSG(c != EOF, "truncated image header");
    if (std::isspace(c)) { in.get(); continue; }
    if (c == '#') { std::string skip; std::getline(in, skip); continue; }
    in >> tokens[i++];
  }
  in.get();   // the single whitespace after maxval
  Pnm image;
  VULCAN_ASSERT_MSG(tokens[0] == "P5" || tokens[0] == "P6", "only binary PGM / PPM files are supported");
  image.channels = tokens[0] == "P5" ? 1 : 3;
  image.width = std::atoi(tokens[1].c_str());
  image.height = std::atoi(tokens[2].c_str());
  image.maxval = std::atoi(tokens[3].c_str());
  VULCAN_ASSERT_MSG(image.width > 0 && image.height > 0 && image.maxval > 0 && image.maxval < 65536, "bad image header");
  const size_t count = (size_t)image.width * image.height * image.channels;
  image.pixels.resize(count);
  const int bytes = image.maxval > 255 ? 2 : 1;
  std::vector<unsigned char> raw(count * bytes);
  in.read(reinterpret_cast<char*>(raw.data()), raw.size());
  VULCAN_ASSERT_MSG((size_t)in.gcount() == raw.size(), "truncated image data");
  for (size_t i = 0; i < count; ++i)
    image.pixels[i] = bytes == 2 ? (unsigned short)((raw[2 * i] << 8) | raw[2 * i + 1]) : raw[i];   // big endian
  return image;
}

void WritePnm(const std::string& file, int width, int height, int channels, int bits, const std::vector<unsigned short>& pixels)
{
  std::ofstream out(file, std::ios::binary);
  VULCAN_ASSERT_MSG(out.is_open(), "unable to write file");
  out << (channels == 1 ? "P5" : "P6") << "\n" << width << " " << height << "\n" << (bits == 16 ? 65535 : 255) << "\n";
  std::vector<unsigned char> raw;
  raw.reserve(pixels.size() * (bits / 8));
  for (unsigned short p : pixels)
  {
    if (bits == 16) raw.push_back((unsigned char)(p >> 8));
    raw.push_back((unsigned char)(p & 0xff));
  }
  out.write(reinterpret_cast<const char*>(raw.data()), raw.size());
}

// cv::Mat::convertTo(type, alpha, beta): saturate_cast<T>(cvRound(v * alpha + beta)), round half to even
unsigned short ConvertPixel(float v, double alpha, double beta, int bits)
{
  const double r = std::nearbyint((double)v * alpha + beta);
  const double hi = bits == 16 ? 65535.0 : 255.0;
  if (!(r > 0)) return 0;
  return (unsigned short)(r > hi ? hi : r);
}

} // namespace

void Image::Load(const std::string& file, float scale)
{
  const Pnm image = ReadPnm(file);
  std::vector<float> host((size_t)image.width * image.height);
  for (size_t i = 0; i < host.size(); ++i)
  {
    float value;
    if (image.channels == 1) value = image.pixels[i];
    else value = (float)std::nearbyint(0.299 * image.pixels[3 * i] + 0.587 * image.pixels[3 * i + 1] + 0.114 * image.pixels[3 * i + 2]);
    host[i] = value * scale;
  }
  Resize(image.width, image.height);
  CopyFromHost(host.data());
}

void Image::Save(const std::string& file, int bits, float alpha, float beta) const
{
  VULCAN_ASSERT_MSG(bits == 8 || bits == 16, "8 or 16 bits");
  std::vector<float> host(GetTotal());
  CopyToHost(host.data());
  std::vector<unsigned short> pixels(host.size());
  for (size_t i = 0; i < host.size(); ++i) pixels[i] = ConvertPixel(host[i], alpha, beta, bits);
  WritePnm(file, GetWidth(), GetHeight(), 1, bits, pixels);
}

void ColorImage::Load(const std::string& file, float scale)
{
  const Pnm image = ReadPnm(file);
  std::vector<Vector3f> host((size_t)image.width * image.height);
  for (size_t i = 0; i < host.size(); ++i)
    for (int c = 0; c < 3; ++c)
      host[i][c] = (float)image.pixels[image.channels == 3 ? 3 * i + c : i] * scale;
  Resize(image.width, image.height);
  CopyFromHost(host.data());
}

void ColorImage::Save(const std::string& file, int bits, float alpha, float beta) const
{
  VULCAN_ASSERT_MSG(bits == 8 || bits == 16, "8 or 16 bits");
  std::vector<Vector3f> host(GetTotal());
  CopyToHost(host.data());
  std::vector<unsigned short> pixels(3 * host.size());
  for (size_t i = 0; i < host.size(); ++i)
    for (int c = 0; c < 3; ++c) pixels[3 * i + c] = ConvertPixel(host[i][c], alpha, beta, bits);
  WritePnm(file, GetWidth(), GetHeight(), 3, bits, pixels);
}

// ---- sequences -------------------------------------------------------------------

namespace
{

std::string FrameFile(const std::string& directory, const char* kind, int index, const char* extension)
{
  char name[64];
  std::snprintf(name, sizeof(name), "/%s_%06d.%s", kind, index, extension);
  return directory + name;
}

std::string Nine(float v)
{
  char text[32];
  std::snprintf(text, sizeof(text), "%.9g", (double)v);
  return text;
}

} // namespace

SequenceWriter::SequenceWriter(const std::string& directory, int width, int height, const Projection& depth_projection,
    const Projection& color_projection, float depth_scale) :
  directory_(directory),
  depth_scale_(depth_scale),
  count_(0),
  closed_(false)
{
  std::ostringstream size;
  size << "size " << width << " " << height;
  lines_.push_back("vulcan-sequence 1");
  lines_.push_back(size.str());
  lines_.push_back("depth_scale " + Nine(depth_scale));
  const Projection* ks[2] = {&depth_projection, &color_projection};
  const char* names[2] = {"depth_projection", "color_projection"};
  for (int i = 0; i < 2; ++i)
    lines_.push_back(std::string(names[i]) + " " + Nine(ks[i]->GetFocalLength()[0]) + " " + Nine(ks[i]->GetFocalLength()[1]) +
        " " + Nine(ks[i]->GetCenterPoint()[0]) + " " + Nine(ks[i]->GetCenterPoint()[1]));
}

SequenceWriter::~SequenceWriter() { if (!closed_) Close(); }

void SequenceWriter::Append(const Frame& frame)
{
  VULCAN_ASSERT_MSG(frame.depth_image, "frame missing depth image");
  frame.depth_image->Save(FrameFile(directory_, "depth", count_, "pgm"), 16, 1.0f / depth_scale_, 0);
  if (frame.color_image) frame.color_image->Save(FrameFile(directory_, "color", count_, "ppm"), 8, 255.0f, 0);
  std::ostringstream pose;
  pose << "pose " << count_;
  const Matrix4f* both[2] = {&frame.depth_to_world_transform.GetMatrix(), &frame.depth_to_world_transform.GetInverseMatrix()};
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) pose << " " << Nine((*both[k])(r, c));
  lines_.push_back(pose.str());
  ++count_;
}

void SequenceWriter::Close()
{
  std::ofstream out(directory_ + "/sequence.txt");
  VULCAN_ASSERT_MSG(out.is_open(), "unable to write sequence.txt");
  for (size_t i = 0; i < 5 && i < lines_.size(); ++i) out << lines_[i] << "\n";
  out << "frames " << count_ << "\n";
  for (size_t i = 5; i < lines_.size(); ++i) out << lines_[i] << "\n";
  closed_ = true;
}

SequenceReader::SequenceReader(const std::string& directory) :
  directory_(directory), width_(0), height_(0), count_(0), depth_scale_(0.001f)
{
  std::ifstream in(directory + "/sequence.txt");
  VULCAN_ASSERT_MSG(in.is_open(), "unable to read sequence.txt");
  std::string line;
  while (std::getline(in, line))
  {
    std::istringstream row(line);
    std::string key;
    row >> key;
    if (key == "size") row >> width_ >> height_;
    else if (key == "depth_scale") row >> depth_scale_;
    else if (key == "frames") { row >> count_; poses_.resize(count_); }
    else if (key == "depth_projection" || key == "color_projection")
    {
      float fx, fy, cx, cy;
      row >> fx >> fy >> cx >> cy;
      Projection& k = key == "depth_projection" ? depth_projection_ : color_projection_;
      k.SetFocalLength(fx, fy);
      k.SetCenterPoint(cx, cy);
    }
    else if (key == "pose")
    {
      int index;
      row >> index;
      vk_transform t;
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) row >> t.m[c * 4 + r];     // file is row-major
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) row >> t.inv[c * 4 + r];
      if (index >= (int)poses_.size()) poses_.resize(index + 1);
      poses_[index] = Transform::FromVk(t);
    }
  }
}

void SequenceReader::Read(int index, Frame& frame) const
{
  VULCAN_ASSERT_MSG(index >= 0 && index < count_, "frame index out of range");
  if (!frame.depth_image) frame.depth_image = std::make_shared<Image>();
  frame.depth_image->Load(FrameFile(directory_, "depth", index, "pgm"), depth_scale_);
  const std::string color = FrameFile(directory_, "color", index, "ppm");
  if (std::ifstream(color).good())
  {
    if (!frame.color_image) frame.color_image = std::make_shared<ColorImage>();
    frame.color_image->Load(color, 1.0f / 255.0f);
  }
  frame.depth_projection = depth_projection_;
  frame.color_projection = color_projection_;
  frame.depth_to_world_transform = index < (int)poses_.size() ? poses_[index] : Transform();
}

} // namespace vulcan
