// host_tests.cpp — the reference's gtest cases for the hot path, re-authored
// against the current Frame fields (depth_to_world_transform, depth_projection)
// and run through the C++ class layer -> C ABI -> HIP kernels. gtest is not
// available in this image, so a few macros stand in for it. Each case names the
// reference test it restates; inputs and tolerances are the reference's.
//
//   ./host_tests            run everything (needs a GPU)
//   ./host_tests <filter>   run the cases whose name contains <filter>
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include <vulcan/vulcan.h>
#include <vulcan/frame.cuh>
#include <vulcan/tracer.cuh>

using namespace vulcan;

// ---- minimal test harness -------------------------------------------------------

struct Failure { std::string text; };

#define STR2(x) #x
#define STR(x) STR2(x)
#define FAIL_HERE(msg) throw Failure{std::string(__FILE__ ":" STR(__LINE__) ": ") + (msg)}
#define ASSERT_TRUE(c) do { if (!(c)) FAIL_HERE("expected true: " #c); } while (0)
#define ASSERT_FALSE(c) do { if (c) FAIL_HERE("expected false: " #c); } while (0)
#define ASSERT_EQ(a, b) do { if (!((a) == (b))) FAIL_HERE("expected equal: " #a " vs " #b + \
    (" (" + std::to_string((double)(a)) + " vs " + std::to_string((double)(b)) + ")")); } while (0)
#define ASSERT_NEAR(a, b, eps) do { const double d__ = std::fabs((double)(a) - (double)(b)); \
    if (!(d__ <= (eps))) FAIL_HERE("expected |" #a " - " #b "| <= " #eps + (", got " + std::to_string(d__) + \
    " (" + std::to_string((double)(a)) + " vs " + std::to_string((double)(b)) + ")")); } while (0)
#define ASSERT_FLOAT_EQ(a, b) ASSERT_NEAR(a, b, 4 * FLT_EPSILON * std::max(std::fabs((double)(a)), std::fabs((double)(b))))
#define ASSERT_THROW(stmt) do { bool t__ = false; try { stmt; } catch (const Exception&) { t__ = true; } \
    if (!t__) FAIL_HERE("expected vulcan::Exception: " #stmt); } while (0)

struct TestCase { const char* name; std::function<void()> body; };
static std::vector<TestCase>& Registry() { static std::vector<TestCase> r; return r; }
struct Registrar { Registrar(const char* n, std::function<void()> f) { Registry().push_back({n, f}); } };
#define TEST(suite, name) static void suite##_##name(); \
    static Registrar reg_##suite##_##name(#suite "." #name, suite##_##name); static void suite##_##name()

template <typename T> std::vector<T> Download(const Buffer<T>& b, size_t n)
{
  std::vector<T> host(n);
  if (n) VK_ASSERT(vk_memcpy_d2h(host.data(), b.GetData(), sizeof(T) * n, Device::GetStream()));
  return host;
}
template <typename T> std::vector<T> Download(const Buffer<T>& b) { return Download(b, b.GetSize()); }
template <typename T> void Upload(Buffer<T>& b, const std::vector<T>& host)
{
  VK_ASSERT(vk_memcpy_h2d(b.GetData(), host.data(), sizeof(T) * host.size(), Device::GetStream()));
}
template <typename T> void UploadAt(T* device, const T& value)
{
  VK_ASSERT(vk_memcpy_h2d(device, &value, sizeof(T), Device::GetStream()));
}

static std::shared_ptr<Image> MakeDepth(int w, int h, const std::function<float(int, int)>& f)
{
  std::vector<float> host(size_t(w) * h);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) host[size_t(y) * w + x] = f(x, y);
  auto image = std::make_shared<Image>(w, h);
  image->CopyFromHost(host.data());
  return image;
}
static std::shared_ptr<ColorImage> MakeColor(int w, int h, const std::function<Vector3f(int, int)>& f)
{
  std::vector<Vector3f> host(size_t(w) * h);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) host[size_t(y) * w + x] = f(x, y);
  auto image = std::make_shared<ColorImage>(w, h);
  image->CopyFromHost(host.data());
  return image;
}
static std::vector<float> Download(const Image& image)
{
  std::vector<float> host(image.GetTotal());
  image.CopyToHost(host.data());
  return host;
}
static std::vector<Vector3f> Download(const ColorImage& image)
{
  std::vector<Vector3f> host(image.GetTotal());
  image.CopyToHost(host.data());
  return host;
}

// ---- layout / math (block_test.cpp, hash_test.cpp, voxel_test.cpp, matrix/transform/projection tests)

TEST(Block, Layout) { ASSERT_EQ(8, sizeof(Block)); ASSERT_EQ(8, Block::resolution); ASSERT_EQ(512, Block::voxel_count); }

TEST(HashEntry, Constructor)
{
  ASSERT_EQ(16, sizeof(HashEntry));
  HashEntry entry;
  ASSERT_FALSE(entry.IsAllocated());
  ASSERT_FALSE(entry.HasNext());
  entry.data = 3; entry.next = 7;
  ASSERT_TRUE(entry.IsAllocated() && entry.HasNext());
  entry.InvalidateData(); entry.InvalidateNext();
  ASSERT_EQ(HashEntry::invalid, entry.data);
  ASSERT_EQ(HashEntry::invalid, entry.next);
}

TEST(Voxel, Empty)
{
  ASSERT_EQ(20, sizeof(Voxel));
  const Voxel v = Voxel::Empty();
  ASSERT_EQ(1.0f, v.distance);
  ASSERT_TRUE(v.GetColor() == Vector3f(0, 0, 0));
  ASSERT_EQ(0, v.distance_weight);
  ASSERT_EQ(0, v.color_weight);
}

TEST(Math, KnownAnswers)
{
  // tests/golden/reference_kats.json (values from the reference headers)
  const Vector2f uv = Projection().Project(0.1f, -0.2f, 1.5f);
  ASSERT_EQ(353.333344f, uv[0]);
  ASSERT_EQ(173.333328f, uv[1]);
  ASSERT_EQ(16, sizeof(Projection)); ASSERT_EQ(128, sizeof(Transform)); ASSERT_EQ(16, sizeof(Light)); ASSERT_EQ(24, sizeof(Vector6f));
  const Transform T = Transform::Translate(0.3f, -1.3f, 3.7f) * Transform::Rotate(0.7474f, 0.3438f, -0.3884f, 0.4152f);
  const Matrix4f I = T.GetMatrix() * T.GetInverseMatrix();
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ASSERT_NEAR(r == c ? 1 : 0, I(r, c), 2e-3);
  const Vector4f p = T.Inverse() * (T * Vector4f(1, 2, 3, 1));
  ASSERT_NEAR(1, p[0], 1e-2); ASSERT_NEAR(2, p[1], 1e-2); ASSERT_NEAR(3, p[2], 1e-2);
  ASSERT_EQ(3, GetKernelBlocks(1025, 512));
}

TEST(Transform, InverseSwapsTheTwoMatrices)   // transform_test.cpp:26-49
{
  const Transform a = Transform::Translate(0.1f, -0.3f, 0.9f) * Transform::Rotate(0.5925f, 0.7040f, -0.2221f, 0.3226f);
  const Transform b = a.Inverse();
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
    {
      ASSERT_FLOAT_EQ(a.GetInverseMatrix()(j, i), b.GetMatrix()(j, i));
      ASSERT_FLOAT_EQ(a.GetMatrix()(j, i), b.GetInverseMatrix()(j, i));
    }
}

TEST(Transform, RotateMatrix)   // transform_test.cpp:51-85
{
  Matrix3f expected;
  expected(0, 0) =  0.6932f; expected(1, 0) = -0.6950f; expected(2, 0) =  0.1910f;
  expected(0, 1) =  0.0696f; expected(1, 1) = -0.1992f; expected(2, 1) = -0.9775f;
  expected(0, 2) =  0.7174f; expected(1, 2) =  0.6909f; expected(2, 2) = -0.0898f;
  const Transform transform = Transform::Rotate(expected);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
    {
      ASSERT_NEAR(expected(j, i), transform.GetMatrix()(j, i), 1E-4);
      ASSERT_NEAR(expected(j, i), transform.GetInverseMatrix()(i, j), 1E-4);
    }
  const Vector3f t = transform.GetTranslation();
  ASSERT_EQ(0, t[0]); ASSERT_EQ(0, t[1]); ASSERT_EQ(0, t[2]);
}

TEST(Transform, Translate)   // transform_test.cpp:132-180
{
  const Vector3f expected(1.1f, -2.7f, 3.9f);
  Transform transform = Transform::Translate(expected);
  Vector3f found = transform.GetTranslation();
  for (int i = 0; i < 3; ++i) ASSERT_FLOAT_EQ(expected[i], found[i]);
  found = Vector3f(transform * Vector4f(0, 0, 0, 1));
  for (int i = 0; i < 3; ++i) ASSERT_FLOAT_EQ(expected[i], found[i]);
  transform = transform.Inverse();
  found = transform.GetTranslation();
  for (int i = 0; i < 3; ++i) ASSERT_FLOAT_EQ(-expected[i], found[i]);
  found = Vector3f(transform * Vector4f(0, 0, 0, 1));
  for (int i = 0; i < 3; ++i) ASSERT_FLOAT_EQ(-expected[i], found[i]);
  transform = Transform::Translate(expected[0], expected[1], expected[2]);
  found = Vector3f(transform * Vector4f(0, 0, 0, 1));
  for (int i = 0; i < 3; ++i) ASSERT_FLOAT_EQ(expected[i], found[i]);
}

TEST(Projection, ProjectAndUnproject)   // projection_test.cpp:58-140
{
  Projection projection;
  projection.SetFocalLength(325, 315);
  projection.SetCenterPoint(325, 235);
  Vector2f found = projection.Project(0, 0, 1);
  ASSERT_FLOAT_EQ(325, found[0]); ASSERT_FLOAT_EQ(235, found[1]);
  const Vector3f point(23.4f, -0.725f, 31.2f);
  found = projection.Project(point);
  ASSERT_FLOAT_EQ((325 * point[0] + 325 * point[2]) / point[2], found[0]);
  ASSERT_FLOAT_EQ((315 * point[1] + 235 * point[2]) / point[2], found[1]);
  Vector3f ray = projection.Unproject(projection.GetCenterPoint());
  ASSERT_FLOAT_EQ(0, ray[0]); ASSERT_FLOAT_EQ(0, ray[1]); ASSERT_FLOAT_EQ(1, ray[2]);
  ray = projection.Unproject(projection.GetCenterPoint(), 17.4f);
  ASSERT_FLOAT_EQ(0, ray[0]); ASSERT_FLOAT_EQ(0, ray[1]); ASSERT_FLOAT_EQ(17.4f, ray[2]);
  const Vector2f uv(112.4f, 401.3f);                       // K^-1 * (u, v, 1) * depth
  ray = projection.Unproject(uv, 2.5f);
  ASSERT_FLOAT_EQ(2.5f * (uv[0] - 325) / 325, ray[0]);
  ASSERT_FLOAT_EQ(2.5f * (uv[1] - 235) / 315, ray[1]);
  ASSERT_FLOAT_EQ(2.5f, ray[2]);
  const Vector2f back = projection.Project(ray);            // round trip
  ASSERT_NEAR(uv[0], back[0], 1e-3); ASSERT_NEAR(uv[1], back[1], 1e-3);
}

TEST(Vector, NormsAndDot)   // matrix_test.cpp:60-230
{
  const Vector3f a(1.5f, -2.0f, 0.5f), b(0.25f, 4.0f, -1.0f);
  ASSERT_FLOAT_EQ(1.5f * 1.5f + 4.0f + 0.25f, a.SquaredNorm());
  ASSERT_FLOAT_EQ(std::sqrt(6.5f), a.Norm());
  ASSERT_FLOAT_EQ(1.5f * 0.25f - 8.0f - 0.5f, a.Dot(b));
  const Vector3f n = a.Normalized();
  ASSERT_NEAR(1.0, n.Norm(), 1e-6);
  for (int i = 0; i < 3; ++i) ASSERT_FLOAT_EQ(a[i] / std::sqrt(6.5f), n[i]);
  Vector3f c = a;
  c.Normalize();
  for (int i = 0; i < 3; ++i) ASSERT_EQ(n[i], c[i]);
  const Vector3f x = Vector3f(1, 0, 0).Cross(Vector3f(0, 1, 0));
  ASSERT_EQ(0, x[0]); ASSERT_EQ(0, x[1]); ASSERT_EQ(1, x[2]);
}

TEST(Exception, What)
{
  try { VULCAN_THROW("boom"); }
  catch (const Exception& e) { ASSERT_TRUE(std::string(e.what()).find("): boom") != std::string::npos); ASSERT_TRUE(e.line() > 0); return; }
  FAIL_HERE("no exception");
}

TEST(Buffer, ResizeReserveCopy)   // buffer_test.cu
{
  Buffer<int> b;
  ASSERT_EQ(0, b.GetSize()); ASSERT_TRUE(b.IsEmpty());
  b.Reserve(100);
  ASSERT_EQ(100, b.GetCapacity()); ASSERT_EQ(0, b.GetSize());
  b.Resize(40);
  ASSERT_EQ(40, b.GetSize()); ASSERT_EQ(100, b.GetCapacity());
  std::vector<int> host(40);
  for (int i = 0; i < 40; ++i) host[i] = i * i;
  b.CopyFromHost(host.data());
  Buffer<int> c(40);
  b.CopyToDevice(c.GetData());
  Device::Synchronize();
  ASSERT_TRUE(Download(c) == host);
  b.Resize(400);   // growth discards contents, keeps the size request (buffer.h:64-73)
  ASSERT_EQ(400, b.GetSize()); ASSERT_EQ(400, b.GetCapacity());
}

// ---- Volume (tests/volume_test.cpp; white-box through a subclass, :23-31) -------

static const int MAIN_BLOCK_COUNT = 1024, EXCESS_BLOCK_COUNT = 512, MAX_BLOCK_COUNT = 1536;

struct OpenVolume : public Volume
{
  OpenVolume() : Volume(MAIN_BLOCK_COUNT, EXCESS_BLOCK_COUNT) {}
  using Volume::ResetBlockVisibility; using Volume::UpdateBlockVisibility;
  using Volume::CreateAllocationRequests; using Volume::HandleAllocationRequests;
  using Volume::voxels_; using Volume::hash_entries_; using Volume::free_voxel_blocks_;
  using Volume::allocation_types_; using Volume::allocation_blocks_; using Volume::block_visibility_;
  using Volume::visible_blocks_; using Volume::max_block_count_; using Volume::empty_; using Volume::voxel_length_;
};

TEST(Volume, Constructor)   // volume_test.cpp:35-98 (trunc default is 0.04, volume.cu:375)
{
  OpenVolume v;
  ASSERT_FLOAT_EQ(0.008f, v.GetVoxelLength());
  ASSERT_FLOAT_EQ(0.04f, v.GetTruncationLength());
  ASSERT_EQ(MAX_BLOCK_COUNT, v.max_block_count_);
  ASSERT_EQ(size_t(MAX_BLOCK_COUNT) * Block::voxel_count, v.voxels_.GetSize());
  ASSERT_EQ(MAX_BLOCK_COUNT, v.hash_entries_.GetSize());
  ASSERT_EQ(MAX_BLOCK_COUNT, v.free_voxel_blocks_.GetSize());
  ASSERT_EQ(MAIN_BLOCK_COUNT, v.allocation_types_.GetSize());
  ASSERT_EQ(MAIN_BLOCK_COUNT, v.allocation_blocks_.GetSize());
  ASSERT_EQ(MAX_BLOCK_COUNT, v.block_visibility_.GetSize());
  ASSERT_EQ(MAX_BLOCK_COUNT, v.visible_blocks_.GetCapacity());
  ASSERT_EQ(0, v.GetVisibleBlocks().GetSize());
  ASSERT_TRUE(v.empty_);
  for (const HashEntry& e : Download(v.hash_entries_)) { ASSERT_TRUE(e.block == Block()); ASSERT_EQ(-1, e.data); ASSERT_EQ(-1, e.next); }
  for (Visibility s : Download(v.block_visibility_)) ASSERT_EQ(VISIBILITY_FALSE, s);
  for (AllocationType t : Download(v.allocation_types_)) ASSERT_EQ(ALLOC_TYPE_NONE, t);
  const std::vector<int> free_list = Download(v.free_voxel_blocks_);
  for (size_t i = 0; i < free_list.size(); ++i) ASSERT_EQ((int)i, free_list[i]);
  const std::vector<Voxel> voxels = Download(v.voxels_, 4096);
  for (const Voxel& x : voxels) { ASSERT_EQ(1.0f, x.distance); ASSERT_EQ(0, x.distance_weight); ASSERT_EQ(0, x.color_weight); }
}

TEST(Volume, ResetBlockVisibility)   // volume_test.cpp:100-122
{
  OpenVolume v;
  std::vector<Visibility> expected(MAX_BLOCK_COUNT);
  for (size_t i = 0; i < expected.size(); ++i) expected[i] = (i % 7 == 0) ? VISIBILITY_TRUE : VISIBILITY_FALSE;
  Upload(v.block_visibility_, expected);
  v.ResetBlockVisibility();
  const std::vector<Visibility> found = Download(v.block_visibility_);
  for (size_t i = 0; i < expected.size(); ++i)
    ASSERT_EQ(expected[i] == VISIBILITY_TRUE ? VISIBILITY_UNKNOWN : VISIBILITY_FALSE, found[i]);
}

TEST(Volume, UpdateBlockVisibility)   // volume_test.cpp:124-248
{
  OpenVolume v;
  Frame frame;
  frame.depth_image = std::make_shared<Image>(640, 480);
  frame.depth_projection.SetFocalLength(320, 320);
  frame.depth_projection.SetCenterPoint(320, 240);
  frame.depth_to_world_transform = Transform::Translate(10, -2, 30);

  auto check = [&](std::vector<int> expected)
  {
    v.UpdateBlockVisibility(frame);
    std::vector<int> found = Download(v.GetVisibleBlocks());
    std::sort(found.begin(), found.end());
    std::sort(expected.begin(), expected.end());
    ASSERT_TRUE(found == expected);
  };

  check({});
  for (int i : {7, 32, 123}) UploadAt(v.block_visibility_.GetData() + i, VISIBILITY_TRUE);
  check({7, 32, 123});

  for (int i : {3, 17, 315}) UploadAt(v.block_visibility_.GetData() + i, VISIBILITY_UNKNOWN);
  const float scale = 1.0f / (Block::resolution * v.voxel_length_);
  HashEntry entry;
  entry.block = Block(10 * scale, -2 * scale, 33 * scale);   UploadAt(v.hash_entries_.GetData() + 3, entry);    // in view
  entry.block = Block(-10 * scale, -2 * scale, 28 * scale);  UploadAt(v.hash_entries_.GetData() + 17, entry);   // behind the camera
  entry.block = Block(11 * scale, -1 * scale, 53 * scale);   UploadAt(v.hash_entries_.GetData() + 315, entry);  // in view
  check({3, 7, 32, 123, 315});
}

static uint32_t HashOf(int bx, int by, int bz, uint32_t K)
{
  return ((bx * 73856093u) ^ (by * 19349669u) ^ (bz * 83492791u)) % K;
}

TEST(Volume, CreateAllocationRequests)   // volume_test.cpp:250-431
{
  OpenVolume v;
  const int w = 64, h = 48;
  const float trunc_length = 0.20f, voxel_length = 0.02f;
  const float block_length = Block::resolution * voxel_length, inv_block_length = 1 / block_length;
  v.SetTruncationLength(trunc_length);
  v.SetVoxelLength(voxel_length);

  Frame frame;
  frame.depth_to_world_transform = Transform::Translate(-10.73f, 2.11f, -33.54f);
  frame.depth_projection.SetFocalLength(32, 32);
  frame.depth_projection.SetCenterPoint(32, 24);
  frame.depth_image = MakeDepth(w, h, [](int x, int y) { return float(1 + 3 * (((x + y) % 100) / 99.0)); });
  const std::vector<float> depth = Download(*frame.depth_image);
  const Transform& Twd = frame.depth_to_world_transform;

  std::vector<AllocationType> exp_types(MAIN_BLOCK_COUNT, ALLOC_TYPE_NONE);
  std::vector<Visibility> exp_visibility(MAX_BLOCK_COUNT, VISIBILITY_FALSE);

  // independent walk along every ray: step the parameter t from block boundary to block boundary
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
    {
      const Vector3f Xcp = depth[y * w + x] * frame.depth_projection.Unproject(Vector2f(x + 0.5f, y + 0.5f));
      const Vector3f Xwp = Vector3f(Twd * Vector4f(Xcp, 1));
      const Vector3f dir = (Xwp - Twd.GetTranslation()).Normalized();
      const Vector3f origin = Xwp - trunc_length * dir;

      for (float t = 0; t < 2 * trunc_length + 1E-6f; )
      {
        const Vector3f current = origin + t * dir;
        int b[3];
        for (int a = 0; a < 3; ++a) b[a] = (int)floorf(current[a] * inv_block_length);
        const uint32_t code = HashOf(b[0], b[1], b[2], MAIN_BLOCK_COUNT);
        exp_types[code] = ALLOC_TYPE_MAIN;
        exp_visibility[code] = VISIBILITY_TRUE;

        float rate[3];
        for (int a = 0; a < 3; ++a)
        {
          const int step = (dir[a] > 0) ? +1 : -1;
          rate[a] = (block_length * (b[a] + std::max(0, step)) - current[a]) / dir[a];
          if (rate[a] < 1E-8f) rate[a] = 1E-6f;
        }
        t += (rate[0] < rate[1]) ? (rate[0] < rate[2] ? rate[0] : rate[2]) : (rate[1] < rate[2] ? rate[1] : rate[2]);
      }
    }

  // occupy the first requested bucket with a foreign block: its requests must become EXCESS
  const size_t first = std::find(exp_types.begin(), exp_types.end(), ALLOC_TYPE_MAIN) - exp_types.begin();
  exp_visibility[first] = VISIBILITY_FALSE;
  exp_types[first] = ALLOC_TYPE_EXCESS;
  HashEntry occupied;
  occupied.block = Block(-1, -1, -1);
  occupied.data = 0;
  UploadAt(v.hash_entries_.GetData() + first, occupied);

  v.CreateAllocationRequests(frame);
  ASSERT_TRUE(Download(v.allocation_types_) == exp_types);
  ASSERT_TRUE(Download(v.block_visibility_) == exp_visibility);
}

TEST(Volume, HandleAllocationRequests)   // volume_test.cpp:433-556
{
  OpenVolume v;
  auto request = [&](int bucket, Block block, AllocationType type)
  {
    UploadAt(v.allocation_blocks_.GetData() + bucket, block);
    UploadAt(v.allocation_types_.GetData() + bucket, type);
  };
  request(0, Block(1, 2, 3), ALLOC_TYPE_MAIN);
  request(323, Block(7, 3, -1), ALLOC_TYPE_MAIN);
  v.HandleAllocationRequests();
  for (AllocationType t : Download(v.allocation_types_)) ASSERT_EQ(ALLOC_TYPE_NONE, t);
  std::vector<HashEntry> e = Download(v.hash_entries_);
  ASSERT_TRUE(e[0].block == Block(1, 2, 3) && !e[0].HasNext());
  ASSERT_TRUE(e[323].block == Block(7, 3, -1) && !e[323].HasNext());
  // the reference accepts either order of the two pool slots; ordered allocation fixes it
  ASSERT_EQ(MAX_BLOCK_COUNT - 1, e[0].data);
  ASSERT_EQ(MAX_BLOCK_COUNT - 2, e[323].data);

  request(0, Block(7, 3, 0), ALLOC_TYPE_EXCESS);
  request(323, Block(-9, 1, -2), ALLOC_TYPE_EXCESS);
  v.HandleAllocationRequests();
  for (AllocationType t : Download(v.allocation_types_)) ASSERT_EQ(ALLOC_TYPE_NONE, t);
  e = Download(v.hash_entries_);
  ASSERT_EQ(MAIN_BLOCK_COUNT + 0, e[0].next);
  ASSERT_EQ(MAIN_BLOCK_COUNT + 1, e[323].next);
  ASSERT_TRUE(e[e[0].next].block == Block(7, 3, 0) && !e[e[0].next].HasNext());
  ASSERT_TRUE(e[e[323].next].block == Block(-9, 1, -2) && !e[e[323].next].HasNext());
  ASSERT_EQ(MAX_BLOCK_COUNT - 3, e[e[0].next].data);
  ASSERT_EQ(MAX_BLOCK_COUNT - 4, e[e[323].next].data);
}

// ---- Integrator (tests/integrator_test.cu) ---------------------------------------

TEST(Integrator, Constructor)   // integrator_test.cu:40-62
{
  auto volume = std::make_shared<Volume>(512, 256);
  ColorIntegrator integrator(volume);
  ASSERT_TRUE(volume == integrator.GetVolume());
  ASSERT_EQ(16, integrator.GetMaxDistanceWeight());
  integrator.SetMaxDistanceWeight(10);
  ASSERT_EQ(10, integrator.GetMaxDistanceWeight());
#ifndef NDEBUG
  ASSERT_THROW(integrator.SetMaxDistanceWeight(0));
  ASSERT_THROW(integrator.SetMaxDistanceWeight(-1));
#endif
}

TEST(Integrator, Integrate)   // integrator_test.cu:82-221
{
  const int w = 160, h = 120;
  const float trunc_length = 0.02f, voxel_length = 0.008f, block_length = Block::resolution * voxel_length;

  Frame frame;
  frame.depth_projection.SetFocalLength(80, 80);
  frame.depth_projection.SetCenterPoint(80, 60);
  frame.color_projection = frame.depth_projection;
  frame.depth_image = MakeDepth(w, h, [](int, int) { return 1.5f; });
  frame.color_image = MakeColor(w, h, [](int, int) { return Vector3f(1, 2, 3); });

  auto volume = std::make_shared<Volume>(4096, 2048);
  volume->SetTruncationLength(trunc_length);
  volume->SetVoxelLength(voxel_length);
  volume->SetView(frame);

  ColorIntegrator integrator(volume);
  integrator.SetMaxDistanceWeight(16);
  integrator.Integrate(frame);

  const std::vector<float> depths = Download(*frame.depth_image);
  const std::vector<int> visible = Download(volume->GetVisibleBlocks());
  const std::vector<HashEntry> entries = Download(volume->GetHashEntries());
  std::vector<Voxel> found = Download(volume->GetVoxels());
  std::vector<Voxel> expected(found.size(), Voxel::Empty());
  std::vector<bool> border(found.size(), false);
  ASSERT_TRUE(visible.size() > 100);

  for (int index : visible)
  {
    const HashEntry& entry = entries[index];
    ASSERT_TRUE(entry.IsAllocated());
    const Vector3f block_offset = block_length * Vector3f(entry.block.GetOrigin());

    for (int z = 0; z < Block::resolution; ++z)
      for (int y = 0; y < Block::resolution; ++y)
        for (int x = 0; x < Block::resolution; ++x)
        {
          const Vector3f Xwp = block_offset + voxel_length * (Vector3f(x, y, z) + 0.5f);
          const Vector3f Xcp = Vector3f(frame.depth_to_world_transform.Inverse() * Vector4f(Xwp, 1));
          const Vector2f uv = frame.depth_projection.Project(Xcp);
          const int voxel_index = entry.data * Block::voxel_count + (z * 64 + y * 8 + x);

          if (std::abs(uv[0]) < 1E-6f || std::abs(uv[0] - w) < 1E-6 || std::abs(uv[1]) < 1E-6f || std::abs(uv[1] - h) < 1E-6)
            border[voxel_index] = true;

          if (uv[0] >= 0 && uv[0] < w && uv[1] >= 0 && uv[1] < h)
          {
            const float distance = depths[int(uv[1]) * w + int(uv[0])] - Xcp[2];
            if (distance > -trunc_length)
            {
              Voxel& voxel = expected[voxel_index];
              float weight = voxel.distance_weight + 1;
              voxel.distance = (voxel.distance_weight * voxel.distance + min(1.0f, distance / trunc_length)) / weight;
              voxel.distance_weight = min(integrator.GetMaxDistanceWeight(), weight);
            }
          }
        }
  }

  auto compare = [&](int weight_factor)
  {
    size_t updated = 0;
    for (size_t i = 0; i < expected.size(); ++i)
      if (!border[i] || (expected[i].distance_weight > 0 && found[i].distance_weight > 0))
      {
        ASSERT_NEAR(expected[i].distance, found[i].distance, 1E-5);
        ASSERT_NEAR(weight_factor * expected[i].distance_weight, found[i].distance_weight, 1E-5);
        updated += expected[i].distance_weight > 0;
      }
    ASSERT_TRUE(updated > 10000);
  };
  compare(1);
  integrator.Integrate(frame);   // same distance, weight x 2 (:210-220)
  found = Download(volume->GetVoxels());
  compare(2);
}

// ---- Tracer (tests/tracer_test.cu) -------------------------------------------------

static Transform TracerTestTcw()
{
  return Transform::Translate(0.3f, -1.3f, 3.7f) * Transform::Rotate(0.7474f, 0.3438f, -0.3884f, 0.4152f);
}

TEST(Tracer, ComputePatches)   // tracer_test.cu:22-166
{
  const float block_length = 0.008f, min_depth = 0.1f, max_depth = 5.0f;
  const int image_width = 640, image_height = 480, bounds_width = 80, bounds_height = 60;
  const Transform Tcw = TracerTestTcw();
  Projection projection;
  projection.SetFocalLength(346.723f, 353.914f);
  projection.SetCenterPoint(321.294f, 239.052f);

  const Vector3f points[5] = { Vector3f(320, 240, 2.5f), Vector3f(120, 340, 1.5f), Vector3f(420, 240, 0.6f),
                               Vector3f(-20, -40, 1.0f), Vector3f(720, 580, 1.0f) };
  std::vector<int> indices;
  std::vector<HashEntry> entries;
  std::vector<Patch> expected;

  for (int i = 0; i < 5; ++i)
  {
    const Vector3f Xcp = projection.Unproject(Vector2f(points[i])) * points[i][2];
    const Vector3f Xwp = Vector3f(Tcw.Inverse() * Vector4f(Xcp, 1.0f));
    HashEntry entry;
    for (int a = 0; a < 3; ++a) entry.block[a] = Xwp[a] / block_length;
    entry.next = -1;
    entry.data = 0;
    indices.push_back(i);
    entries.push_back(entry);

    Vector2i bmin(INT_MAX, INT_MAX), bmax(INT_MIN, INT_MIN);
    Vector2f drng(+FLT_MAX, -FLT_MAX);
    for (int c = 0; c < 8; ++c)
    {
      const Vector4f corner(block_length * ((c & 1) + entry.block[0]), block_length * (((c >> 1) & 1) + entry.block[1]),
                            block_length * (((c >> 2) & 1) + entry.block[2]), 1);
      const Vector3f P = Vector3f(Tcw * corner);
      const Vector2f uv = projection.Project(P);
      const float u = bounds_width * (uv[0] / image_width), v = bounds_height * (uv[1] / image_height);
      bmin[0] = clamp(min((int)floorf(u), bmin[0]), 0, bounds_width - 1);
      bmin[1] = clamp(min((int)floorf(v), bmin[1]), 0, bounds_height - 1);
      bmax[0] = clamp(max((int)ceilf(u), bmax[0]), 0, bounds_width - 1);
      bmax[1] = clamp(max((int)ceilf(v), bmax[1]), 0, bounds_height - 1);
      drng[0] = min(P[2], drng[0]);
      drng[1] = max(P[2], drng[1]);
    }
    const int gx = (bmax[0] - bmin[0] + Patch::max_size - 1) / Patch::max_size;
    const int gy = (bmax[1] - bmin[1] + Patch::max_size - 1) / Patch::max_size;
    for (int j = 0; j < gy; ++j)
      for (int k = 0; k < gx; ++k)
      {
        Patch patch;
        patch.bounds = drng;
        patch.origin = Vector2s(bmin[0] + Patch::max_size * k, bmin[1] + Patch::max_size * j);
        patch.size = Vector2s(min(bmax[0] - patch.origin[0] + 1, Patch::max_size), min(bmax[1] - patch.origin[1] + 1, Patch::max_size));
        expected.push_back(patch);
      }
  }

  Buffer<int> d_indices(indices.size()), d_count(1);
  Buffer<HashEntry> d_entries(entries.size());
  Buffer<Patch> d_patches(10 * expected.size());
  Upload(d_indices, indices);
  Upload(d_entries, entries);
  VK_ASSERT(vk_memset(d_count.GetData(), 0, sizeof(int), Device::GetStream()));

  vulcan::ComputePatches(d_indices.GetData(), d_entries.GetData(), Tcw, projection, block_length, min_depth,
      max_depth, indices.size(), image_width, image_height, bounds_width, bounds_height, d_patches.GetData(),
      d_count.GetData());

  ASSERT_EQ((int)expected.size(), Download(d_count)[0]);
  const std::vector<Patch> found = Download(d_patches, expected.size());
  for (const Patch& f : found)
  {
    bool matched = false;
    for (Patch& e : expected)
      if (std::fabs(e.bounds[0] - f.bounds[0]) < 1E-4f && std::fabs(e.bounds[1] - f.bounds[1]) < 1E-4f &&
          e.origin == f.origin && e.size == f.size)
      {
        matched = true;
        e.bounds[0] = NAN;
        break;
      }
    ASSERT_TRUE(matched);
  }
}

TEST(Tracer, ComputeBounds)   // tracer_test.cu:168-244
{
  const int bounds_width = 80, bounds_height = 60;
  std::vector<Vector2f> expected(bounds_width * bounds_height, Vector2f(+FLT_MAX, -FLT_MAX));
  std::vector<Patch> patches;
  auto add = [&](Vector2s origin, Vector2s size, Vector2f bounds) { Patch p; p.origin = origin; p.size = size; p.bounds = bounds; patches.push_back(p); };
  add(Vector2s(23, 46), Vector2s(5, 2), Vector2f(1.237f, 1.523f));
  add(Vector2s(3, 9), Vector2s(1, 1), Vector2f(2.021f, 3.214f));
  add(Vector2s(20, 43), Vector2s(5, 8), Vector2f(0.856f, 1.014f));
  add(Vector2s(0, 0), Vector2s(2, 2), Vector2f(1.256f, 2.114f));
  add(Vector2s(79, 59), Vector2s(1, 1), Vector2f(0.256f, 1.314f));
  add(Vector2s(3, 9), Vector2s(3, 3), Vector2f(0.256f, 1.314f));
  for (const Patch& p : patches)
    for (int i = 0; i < p.size[1]; ++i)
      for (int j = 0; j < p.size[0]; ++j)
      {
        Vector2f& cell = expected[(p.origin[1] + i) * bounds_width + p.origin[0] + j];
        cell[0] = min(p.bounds[0], cell[0]);
        cell[1] = max(p.bounds[1], cell[1]);
      }
  Buffer<Patch> d_patches(patches.size());
  Buffer<Vector2f> d_bounds(expected.size());
  Upload(d_patches, patches);
  vulcan::ResetBoundsBuffer(d_bounds.GetData(), d_bounds.GetSize());
  vulcan::ComputeBounds(d_patches.GetData(), d_bounds.GetData(), bounds_width, patches.size());
  const std::vector<Vector2f> found = Download(d_bounds);
  for (size_t i = 0; i < expected.size(); ++i) { ASSERT_FLOAT_EQ(expected[i][0], found[i][0]); ASSERT_FLOAT_EQ(expected[i][1], found[i][1]); }
}

static void FuseToFixedPoint(const std::shared_ptr<Volume>& volume, const Frame& frame)
{
  size_t visible_count = 0;
  do   // tracer_test.cu:298-303
  {
    visible_count = volume->GetVisibleBlocks().GetSize();
    volume->SetView(frame);
  }
  while (visible_count != volume->GetVisibleBlocks().GetSize());
  ColorIntegrator integrator(volume);
  integrator.Integrate(frame);
}

TEST(Tracer, ComputePoints)   // tracer_test.cu:246-390: plane at 1.5 m, rotated pose
{
  const int w = 640, h = 480;
  auto volume = std::make_shared<Volume>(4096, 2048);
  volume->SetTruncationLength(0.04f);
  volume->SetVoxelLength(0.008f);

  Frame frame;
  frame.depth_to_world_transform = TracerTestTcw().Inverse();
  frame.depth_projection.SetFocalLength(546.723f, 553.914f);
  frame.depth_projection.SetCenterPoint(321.294f, 239.052f);
  frame.color_projection = frame.depth_projection;
  frame.depth_image = MakeDepth(w, h, [](int, int) { return 1.5f; });
  frame.color_image = MakeColor(w, h, [](int, int) { return Vector3f(0.1f, 0.2f, 0.3f); });
  FuseToFixedPoint(volume, frame);

  Frame traced;
  traced.depth_to_world_transform = frame.depth_to_world_transform;
  traced.depth_projection = traced.color_projection = frame.depth_projection;
  traced.depth_image = std::make_shared<Image>(w, h);
  Tracer tracer(volume);
  tracer.Trace(traced);

  const std::vector<float> depths = Download(*traced.depth_image);
  const std::vector<Vector3f> colors = Download(*traced.color_image);
  const std::vector<Vector3f> normals = Download(*traced.normal_image);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
    {
      const int i = y * w + x;
      if (!(x <= 2 || x >= w - 2 || y <= 2 || y >= h - 2)) ASSERT_NEAR(1.5f, depths[i], 0.01);
      if (!(x <= 5 || x >= w - 5 || y <= 5 || y >= h - 5))
      {
        ASSERT_NEAR(0.1f, colors[i][0], 0.001); ASSERT_NEAR(0.2f, colors[i][1], 0.001); ASSERT_NEAR(0.3f, colors[i][2], 0.001);
        ASSERT_NEAR(-1.0f, normals[i][2], 0.02);   // dy x dx of a fronto-parallel plane
      }
    }
}

TEST(Tracer, ComputeNormals)   // tracer_test.cu:392-588: sphere cap + checker colour
{
  const int w = 640, h = 480;
  auto volume = std::make_shared<Volume>(2 * 4096, 2 * 2048);
  volume->SetTruncationLength(0.04f);
  volume->SetVoxelLength(0.008f);
  const Vector2f center = 0.5f * Vector2f(w, h);
  auto radius = [&](int x, int y) { return (Vector2f(x + 0.5f, y + 0.5f) - center).Norm(); };

  Frame frame;
  frame.depth_to_world_transform = TracerTestTcw().Inverse();
  frame.depth_projection.SetFocalLength(546.723f, 553.914f);
  frame.depth_projection.SetCenterPoint(321.294f, 239.052f);
  frame.color_projection = frame.depth_projection;
  frame.depth_image = MakeDepth(w, h, [&](int x, int y)
  {
    const float rr = (Vector2f(x + 0.5f, y + 0.5f) - center).SquaredNorm();
    return std::sqrt(rr) < 200 ? float(4.0 - 1.5 * (std::sqrt(200 * 200 - rr) / 200)) : 0.0f;
  });
  frame.color_image = MakeColor(w, h, [&](int x, int y)
  {
    return radius(x, y) < 200 ? Vector3f(0, (x % 40 < 20) ^ (y % 40 < 20), 1) : Vector3f(0, 0, 0);
  });
  const std::vector<float> expected_depths = Download(*frame.depth_image);
  const std::vector<Vector3f> expected_colors = Download(*frame.color_image);
  FuseToFixedPoint(volume, frame);

  Frame traced;
  traced.depth_to_world_transform = frame.depth_to_world_transform;
  traced.depth_projection = traced.color_projection = frame.depth_projection;
  traced.depth_image = std::make_shared<Image>(w, h);
  Tracer tracer(volume);
  tracer.Trace(traced);

  const std::vector<float> depths = Download(*traced.depth_image);
  const std::vector<Vector3f> colors = Download(*traced.color_image);
  const std::vector<Vector3f> normals = Download(*traced.normal_image);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
    {
      const int i = y * w + x;
      const float r = radius(x, y);
      if (180 <= r && r <= 203) continue;
      ASSERT_NEAR(expected_depths[i], depths[i], 0.05);
      if (depths[i] > 0 && r < 170) ASSERT_NEAR(1.0f, normals[i].Norm(), 1e-4);   // upstream: "TODO: compute normal"
      if (x % 20 < 5 || x % 20 > 15 || y % 20 < 5 || y % 20 > 15) continue;
      for (int c = 0; c < 3; ++c) ASSERT_NEAR(expected_colors[i][c], colors[i][c], 0.005);
    }
}

// ---- DepthTracker (tests/depth_tracker_test.cu) ------------------------------------

static void MakeIcpFrame(Frame& frame, bool rippled)
{
  const int w = 640, h = 480;
  frame.depth_projection.SetFocalLength(547, 547);
  frame.depth_projection.SetCenterPoint(320, 240);
  frame.color_projection = frame.depth_projection;
  frame.depth_to_world_transform = rippled
      ? Transform::Translate(0.001f, -0.002f, 0.003f) * Transform::Rotate(0.9998719f, 0.0085884f, -0.0104268f, 0.0085884f)
      : Transform();
  frame.depth_image = MakeDepth(w, h, [&](int x, int y)
  {
    float d = 1;
    if (rippled) { d += 0.01 * cos(16 * M_PI * x / (w - 1)); d += 0.01 * cos(16 * M_PI * y / (h - 1)); }
    return d;
  });
  frame.color_image = MakeColor(w, h, [](int, int) { return Vector3f(0.1f, 0.2f, 0.3f); });
  frame.ComputeNormals();
}

// float64 restatement of the residual (depth_tracker_test.cu:128-204)
static std::vector<double> Residuals64(const Frame& key, const Frame& frm, const Transform& Twc, const Transform& base_Twc)
{
  const int w = frm.depth_image->GetWidth(), h = frm.depth_image->GetHeight();
  const std::vector<float> fd = Download(*frm.depth_image), kd = Download(*key.depth_image);
  const std::vector<Vector3f> fn = Download(*frm.normal_image), kn = Download(*key.normal_image);
  const Vector2f f = key.depth_projection.GetFocalLength(), c = key.depth_projection.GetCenterPoint();
  auto apply = [](const Matrix4f& M, const double* p, double wgt, double* out)
  {
    for (int r = 0; r < 3; ++r) out[r] = (double)M(r, 0) * p[0] + (double)M(r, 1) * p[1] + (double)M(r, 2) * p[2] + (double)M(r, 3) * wgt;
  };
  const Matrix4f Twm = key.depth_to_world_transform.GetMatrix(), Tmw = key.depth_to_world_transform.GetInverseMatrix();
  std::vector<double> residuals(size_t(w) * h, 0.0);

  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
    {
      const int i = y * w + x;
      const double d = fd[i];
      if (!(d > 0)) continue;
      const double Xcp[3] = { d * (x + 0.5 - c[0]) / f[0], d * (y + 0.5 - c[1]) / f[1], d };
      double Xwp[3], bXwp[3], bXmp[3];
      apply(Twc.GetMatrix(), Xcp, 1, Xwp);
      apply(base_Twc.GetMatrix(), Xcp, 1, bXwp);
      apply(Tmw, bXwp, 1, bXmp);
      const double ku = f[0] * bXmp[0] / bXmp[2] + c[0], kv = f[1] * bXmp[1] / bXmp[2] + c[1];
      if (!(ku >= 0 && ku < w && kv >= 0 && kv < h)) continue;
      const int kx = (int)ku, ky = (int)kv, ki = ky * w + kx;
      const double kdepth = kd[ki];
      if (!(kdepth > 0)) continue;
      const double fnl[3] = { fn[i][0], fn[i][1], fn[i][2] }, knl[3] = { kn[ki][0], kn[ki][1], kn[ki][2] };
      double fnw[3], knw[3];
      apply(Twc.GetMatrix(), fnl, 0, fnw);
      apply(Twm, knl, 0, knw);
      const double kk = knw[0] * knw[0] + knw[1] * knw[1] + knw[2] * knw[2];
      const double fk = fnw[0] * knw[0] + fnw[1] * knw[1] + fnw[2] * knw[2];
      if (!(kk > 0 && fk > 0.5)) continue;
      const double Ymp[3] = { kdepth * (kx + 0.5 - c[0]) / f[0], kdepth * (ky + 0.5 - c[1]) / f[1], kdepth };
      double Ywp[3];
      apply(Twm, Ymp, 1, Ywp);
      double bd = 0, r = 0;
      for (int a = 0; a < 3; ++a) { bd += (bXwp[a] - Ywp[a]) * (bXwp[a] - Ywp[a]); r += (Xwp[a] - Ywp[a]) * knw[a]; }
      if (bd < 0.05) residuals[i] = r;
    }
  return residuals;
}

TEST(DepthTracker, Residuals)   // depth_tracker_test.cu:384-422
{
  auto keyframe = std::make_shared<Frame>();
  MakeIcpFrame(*keyframe, false);
  DepthTracker tracker;
  tracker.SetKeyframe(keyframe);
  Buffer<float> buffer;

  Frame same;
  MakeIcpFrame(same, false);
  tracker.ComputeResiduals(same, buffer);
  for (float r : Download(buffer)) ASSERT_EQ(0, r);

  Frame frame;
  MakeIcpFrame(frame, true);
  tracker.ComputeResiduals(frame, buffer);
  const std::vector<float> found = Download(buffer);
  const std::vector<double> expected = Residuals64(*keyframe, frame, frame.depth_to_world_transform, frame.depth_to_world_transform);
  size_t nonzero = 0;
  for (size_t i = 0; i < expected.size(); ++i) { ASSERT_NEAR(expected[i], found[i], 1E-6); nonzero += found[i] != 0; }
  ASSERT_TRUE(nonzero > expected.size() * 9 / 10);
}

TEST(DepthTracker, Jacobian)   // depth_tracker_test.cu:348-382: central differences through ApplyUpdate
{
  auto keyframe = std::make_shared<Frame>();
  MakeIcpFrame(*keyframe, false);
  Frame frame;
  MakeIcpFrame(frame, true);
  DepthTracker tracker;
  tracker.SetTranslationEnabled(true);
  tracker.SetKeyframe(keyframe);
  Buffer<Vector6f> buffer;
  tracker.ComputeJacobian(frame, buffer);
  const std::vector<Vector6f> found = Download(buffer);

  const Transform base = frame.depth_to_world_transform;
  const double steps[6] = { 1E-2, 1E-2, 1E-2, 1E-3, 1E-3, 1E-3 };
  for (int p = 0; p < 6; ++p)
  {
    std::vector<double> side[2];
    for (int s = 0; s < 2; ++s)
    {
      Vector6f update = Vector6f::Zeros();
      update[p] = (s == 0 ? +1 : -1) * steps[p];
      Frame moved = frame;
      moved.depth_to_world_transform = base;
      tracker.ApplyUpdate(moved, update);
      side[s] = Residuals64(*keyframe, frame, moved.depth_to_world_transform, base);
    }
    for (size_t i = 0; i < found.size(); ++i)
      ASSERT_NEAR((side[0][i] - side[1][i]) / (2 * steps[p]), found[i][p], 7E-4);
  }
}

static std::shared_ptr<Frame> CurvedKeyframe()
{
  const int w = 640, h = 480;
  auto key = std::make_shared<Frame>();
  key->depth_projection.SetFocalLength(547, 547);
  key->depth_projection.SetCenterPoint(320, 240);
  key->color_projection = key->depth_projection;
  key->depth_image = MakeDepth(w, h, [&](int x, int y) { return float(1.0 + 0.05 * cos(3.0 * x / w) * sin(2.0 * y / h)); });
  key->color_image = MakeColor(w, h, [](int, int) { return Vector3f(0.1f, 0.2f, 0.3f); });
  key->ComputeNormals();
  return key;
}

TEST(DepthTracker, Track)   // no upstream case; mirrors LightTracker.Track (light_tracker_test.cu:585-669)
{
  auto keyframe = CurvedKeyframe();
  Frame frame = *keyframe;   // same images, slightly wrong pose: must return to the keyframe pose
  frame.depth_to_world_transform = Transform::Translate(0.002f, -0.001f, 0.003f) * Transform::Rotate(0.999995f, 0.002f, -0.0015f, 0.001f);
  DepthTracker tracker;
  tracker.SetKeyframe(keyframe);
  tracker.Track(frame);
  const Matrix4f M = frame.depth_to_world_transform.GetMatrix();
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ASSERT_NEAR(r == c ? 1 : 0, M(r, c), 5e-4);
}

static int g_hook_calls = 0;

static void DoublingHook(float* system_device, int count, void*)   // what a 2-rank all-reduce of equal systems does
{
  std::vector<float> host(count);
  VK_ASSERT(vk_memcpy_d2h(host.data(), system_device, sizeof(float) * count, Device::GetStream()));
  for (float& v : host) v *= 2.0f;
  VK_ASSERT(vk_memcpy_h2d(system_device, host.data(), sizeof(float) * count, Device::GetStream()));
  ++g_hook_calls;
}

TEST(DepthTracker, ReduceHook)   // the hook sits between the sums and the solve
{
  auto keyframe = CurvedKeyframe();
  const Transform start = Transform::Translate(0.002f, -0.001f, 0.003f) * Transform::Rotate(0.999995f, 0.002f, -0.0015f, 0.001f);
  Matrix4f poses[2];
  for (int hooked = 0; hooked < 2; ++hooked)
  {
    Frame frame = *keyframe;
    frame.depth_to_world_transform = start;
    DepthTracker tracker;
    tracker.SetKeyframe(keyframe);
    tracker.SetMaxIterations(5);
    tracker.SetPollChunk(0);      // enqueue every step: one hook call per step, converged or not
    g_hook_calls = 0;
    if (hooked) tracker.SetReduceHook(DoublingHook, nullptr);
    tracker.Track(frame);
    poses[hooked] = frame.depth_to_world_transform.GetMatrix();
    ASSERT_EQ(hooked ? 5 : 0, g_hook_calls);
    if (hooked)
    {
      // default: the loop looks at the convergence mirror every 4 steps and stops enqueuing
      // once |update| < 1e-6 (vk_track_poll) — same pose, no launches after convergence
      Frame again = *keyframe;
      again.depth_to_world_transform = start;
      DepthTracker chunked;
      chunked.SetKeyframe(keyframe);
      chunked.SetMaxIterations(5);
      chunked.SetReduceHook(DoublingHook, nullptr);
      g_hook_calls = 0;
      chunked.Track(again);
      ASSERT_TRUE(g_hook_calls == 4 || g_hook_calls == 5);
      const Matrix4f M = again.depth_to_world_transform.GetMatrix();
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ASSERT_EQ(poses[1](r, c), M(r, c));
    }
  }
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ASSERT_EQ(poses[0](r, c), poses[1](r, c));   // x2 is exact
}

TEST(PyramidTracker, Track)   // pyramid_tracker.cpp:52-90
{
  auto keyframe = CurvedKeyframe();
  Frame frame = *keyframe;
  frame.depth_to_world_transform = Transform::Translate(0.004f, -0.002f, 0.005f) * Transform::Rotate(0.99998f, 0.004f, -0.003f, 0.002f);
  PyramidTracker<DepthTracker> tracker;
  tracker.SetKeyframe(keyframe);
  tracker.Track(frame);
  ASSERT_EQ(20, tracker.GetTracker()->GetMaxIterations());
  const Matrix4f M = frame.depth_to_world_transform.GetMatrix();
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ASSERT_NEAR(r == c ? 1 : 0, M(r, c), 5e-4);
}

TEST(Frame, DownsampleAndFilter)   // frame.cpp:8-58, image.cu:101-165
{
  const int w = 64, h = 48;
  Frame frame;
  frame.depth_image = MakeDepth(w, h, [](int x, int y) { return 1.0f + 0.01f * x + 0.02f * y; });
  frame.color_image = MakeColor(w, h, [](int x, int y) { return Vector3f(x, y, x + y); });
  frame.ComputeNormals();
  Frame half;
  frame.Downsample(half);
  ASSERT_EQ(32, half.depth_image->GetWidth()); ASSERT_EQ(24, half.depth_image->GetHeight());
  ASSERT_FLOAT_EQ(250, half.depth_projection.GetFocalLength()[0]);
  ASSERT_FLOAT_EQ(120, half.depth_projection.GetCenterPoint()[1]);
  const std::vector<float> d = Download(*half.depth_image);
  const std::vector<Vector3f> c = Download(*half.color_image);
  for (int y = 0; y < 24; ++y)
    for (int x = 0; x < 32; ++x)
    {
      ASSERT_EQ(1.0f + 0.01f * (2 * x) + 0.02f * (2 * y), d[y * 32 + x]);   // nearest
      ASSERT_FLOAT_EQ(2 * x + 0.5f, c[y * 32 + x][0]);                       // 2x2 box
      ASSERT_FLOAT_EQ(2 * y + 0.5f, c[y * 32 + x][1]);
    }
  frame.FilterDepths();   // bilateral filter of a linear ramp stays within the ramp's local range
  const std::vector<float> filtered = Download(*frame.depth_image);
  ASSERT_NEAR(1.0f + 0.01f * 30 + 0.02f * 20, filtered[20 * w + 30], 5e-3);
}

// ---- ColorTracker (color_tracker_test.cu) ---------------------------------------------------

static std::shared_ptr<Frame> TexturedFrame(double frequency, bool rippled, const Transform& pose)
{
  const int w = 640, h = 480;   // color_tracker_test.cu:12-148 CreateKeyframeX / CreateFrameX
  auto frame = std::make_shared<Frame>();
  frame->depth_projection.SetFocalLength(547, 547);
  frame->depth_projection.SetCenterPoint(320, 240);
  frame->color_projection = frame->depth_projection;
  frame->depth_to_world_transform = pose;
  frame->depth_image = MakeDepth(w, h, [&](int x, int y)
  {
    float d = 1;
    if (rippled) { d += 0.01 * cos(16 * M_PI * x / (w - 1)); d += 0.01 * cos(16 * M_PI * y / (h - 1)); }
    return d;
  });
  frame->color_image = MakeColor(w, h, [&](int x, int y)
  {
    float c = 0.5f;
    c += 0.245 * cosf(frequency * M_PI * (float(x) / (w - 1)));
    c += 0.245 * cosf(frequency * M_PI * (float(y) / (h - 1)));
    return Vector3f(c, c, c);
  });
  frame->ComputeNormals();
  return frame;
}

static const Transform kColorKeyPose = Transform::Translate(0.0011f, -0.0019f, 0.0031f) * Transform::Rotate(0.9998715f, 0.0086385f, -0.0103759f, 0.0086385f);
static const Transform kColorFramePose = Transform::Translate(0.001f, -0.002f, 0.003f) * Transform::Rotate(0.9998719f, 0.0085884f, -0.0104268f, 0.0085884f);

TEST(ColorTracker, ResidualsOfIdenticalFramesAreZero)   // color_tracker_test.cu:512-521
{
  auto keyframe = TexturedFrame(3.0, false, kColorKeyPose);
  ColorTracker tracker;
  tracker.SetKeyframe(keyframe);
  Buffer<float> buffer;
  tracker.ComputeResiduals(*keyframe, buffer);
  std::vector<float> found(buffer.GetSize());
  buffer.CopyToHost(found.data());
  ASSERT_EQ(size_t(640 * 480), found.size());
  for (float r : found) ASSERT_NEAR(0, r, 1E-6);
}

TEST(ColorTracker, JacobianPredictsResidualChange)   // first-order check through ApplyUpdate
{
  auto keyframe = TexturedFrame(3.0, false, kColorKeyPose);
  auto frame = TexturedFrame(4.0, true, kColorFramePose);
  ColorTracker tracker;
  tracker.SetKeyframe(keyframe);
  Buffer<float> rbuf;
  Buffer<Vector6f> jbuf;
  tracker.ComputeJacobian(*frame, jbuf);
  tracker.ComputeResiduals(*frame, rbuf);
  std::vector<Vector6f> J(jbuf.GetSize());
  std::vector<float> r0(rbuf.GetSize()), r1(rbuf.GetSize());
  jbuf.CopyToHost(J.data());
  rbuf.CopyToHost(r0.data());

  Vector6f step = Vector6f::Zeros();
  step[3] = 1e-3f;   // translate along x
  Frame moved = *frame;
  tracker.ApplyUpdate(moved, step);
  tracker.ComputeResiduals(moved, rbuf);
  rbuf.CopyToHost(r1.data());
  // least-squares slope of the measured change against the predicted one
  double pm = 0, pp = 0;
  int used = 0;
  for (int y = 40; y < 440; ++y)
    for (int x = 40; x < 600; ++x)
    {
      const int i = y * 640 + x;
      if (r0[i] == 0 || r1[i] == 0) continue;
      const double predicted = J[i][3] * step[3];
      pm += predicted * (r1[i] - r0[i]);
      pp += predicted * predicted;
      ++used;
    }
  ASSERT_TRUE(used > 100000);
  ASSERT_NEAR(1.0, pm / pp, 0.1);
}

TEST(ColorTracker, Track)   // no upstream case
{
  // A fronto-parallel plane with a smooth texture leaves rotation and in-plane
  // translation nearly interchangeable, so the pose itself is not pinned; the
  // photometric cost must fall (the CPU restatement reaches 0.08x in 20 steps).
  auto keyframe = TexturedFrame(3.0, false, kColorKeyPose);
  Frame frame = *keyframe;   // same images, slightly wrong pose
  frame.depth_to_world_transform = Transform::Translate(0.004f, -0.002f, 0.001f) * kColorKeyPose;
  ColorTracker tracker;
  tracker.SetKeyframe(keyframe);
  Buffer<float> buffer;
  auto cost = [&](const Frame& f)
  {
    tracker.ComputeResiduals(f, buffer);
    std::vector<float> r(buffer.GetSize());
    buffer.CopyToHost(r.data());
    double sum = 0;
    for (float v : r) sum += double(v) * v;
    return sum;
  };
  const double before = cost(frame);
  tracker.Track(frame);
  ASSERT_TRUE(cost(frame) < 0.2 * before);
  const Matrix4f M = frame.depth_to_world_transform.GetMatrix();
  const Matrix4f I = M * frame.depth_to_world_transform.GetInverseMatrix();
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ASSERT_NEAR(r == c ? 1 : 0, I(r, c), 1e-5);   // still rigid

  PyramidTracker<ColorTracker> pyramid;
  frame.depth_to_world_transform = Transform::Translate(0.004f, -0.002f, 0.001f) * kColorKeyPose;
  pyramid.SetKeyframe(keyframe);
  pyramid.Track(frame);
  ASSERT_TRUE(cost(frame) < 0.3 * before);
}

// ---- LightTracker (light_tracker_test.cu) ---------------------------------------------------

// light_tracker_test.cu:12-189 CreateKeyframeY / CreateFrameY: a camera in front of
// the textured world plane z = 1; `shaded` multiplies the albedo by the light's shading
static std::shared_ptr<Frame> PlaneFrame(const Transform& Twc, const Light& light, bool shaded)
{
  const int w = 640, h = 480;
  auto frame = std::make_shared<Frame>();
  frame->depth_projection.SetFocalLength(547, 547);
  frame->depth_projection.SetCenterPoint(320, 240);
  frame->color_projection = frame->depth_projection;
  frame->depth_to_world_transform = Twc;
  const Vector3f origin = Twc.GetTranslation();
  const Transform Tcw = Twc.Inverse();
  auto world_point = [&](int x, int y)
  {
    const Vector3f Xcp = frame->depth_projection.Unproject(x + 0.5f, y + 0.5f);
    const Vector3f dir = Vector3f(Twc * Vector4f(Xcp, 0));
    const float length = (1 - origin[2]) / dir[2];
    return Vector3f(origin + length * dir);
  };
  frame->depth_image = MakeDepth(w, h, [&](int x, int y) { return Vector3f(Tcw * Vector4f(world_point(x, y), 1))[2]; });
  frame->color_image = MakeColor(w, h, [&](int x, int y)
  {
    const Vector3f Xwp = world_point(x, y);
    const Vector3f Xcp = Vector3f(Tcw * Vector4f(Xwp, 1));
    const Vector3f normal = Vector3f(Tcw * Vector4f(0, 0, -1, 0));
    float c = 0.5f;
    c += 0.245 * cosf(3.0 * M_PI * Xwp[0]);
    c += 0.245 * cosf(3.0 * M_PI * Xwp[1]);
    if (shaded) c *= light.GetShading(Xcp, normal);
    return Vector3f(c, c, c);
  });
  frame->ComputeNormals();
  return frame;
}

static void AssertSamePose(const Transform& a, const Transform& b, double eps)
{
  const Matrix4f diff = a.GetInverseMatrix() * b.GetMatrix();
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ASSERT_NEAR(r == c ? 1 : 0, diff(r, c), eps);
}

TEST(LightTracker, Residuals)   // light_tracker_test.cu:530-561: the shaded keyframe against itself
{
  Light light;
  light.SetIntensity(2.0f);
  light.SetPosition(0.1f, 0.0f, 0.0f);
  const Transform key_pose = Transform::Translate(0.0011f, -0.0019f, -0.5531f) * Transform::Rotate(0.9998715f, 0.0086385f, -0.0103759f, 0.0086385f);
  auto keyframe = PlaneFrame(key_pose, light, false);
  auto same = PlaneFrame(key_pose, light, true);
  LightTracker tracker;
  tracker.SetKeyframe(keyframe);
  tracker.SetLight(light);
  Buffer<float> buffer;
  tracker.ComputeResiduals(*same, buffer);
  std::vector<float> found(buffer.GetSize());
  buffer.CopyToHost(found.data());
  for (float r : found) ASSERT_NEAR(0, r, 1E-4);
}

TEST(LightTracker, Track)   // light_tracker_test.cu:586-669
{
  Light light;
  light.SetIntensity(2.0f);
  light.SetPosition(0.1f, 0.0f, 0.0f);
  const Transform key_pose = Transform::Translate(0.0011f, -0.0019f, -0.5531f) * Transform::Rotate(0.9998715f, 0.0086385f, -0.0103759f, 0.0086385f);
  const Transform frame_pose = Transform::Translate(0.0010f, -0.002f, -0.4030f) * Transform::Rotate(0.9998719f, 0.0085884f, -0.0104268f, 0.0085884f);
  auto keyframe = PlaneFrame(key_pose, light, false);
  auto frame = PlaneFrame(frame_pose, light, true);
  LightTracker tracker;
  tracker.SetKeyframe(keyframe);
  tracker.SetTranslationEnabled(true);
  tracker.SetLight(light);

  for (int i = 0; i < 20; ++i) tracker.Track(*frame);                 // from the true pose: stays
  AssertSamePose(frame_pose, frame->depth_to_world_transform, 1E-5);

  frame->depth_to_world_transform = Transform::Translate(0.1f, 0.1f, 0.1f) *
      Transform::Rotate(0.999871f, 0.008638f, -0.010375f, 0.008638f) * frame_pose;
  for (int i = 0; i < 20; ++i) tracker.Track(*frame);                 // perturbed: comes back
  AssertSamePose(frame_pose, frame->depth_to_world_transform, 1E-5);
}

// ---- Detector (no upstream case: tests/detector_test.cu is empty) -------------------------

TEST(Detector, Constructor)   // detector.cu:66-72, 214-221
{
  Detector detector;
  ASSERT_FLOAT_EQ(2.0f, detector.GetRadius());
  ASSERT_EQ(100, detector.GetMinInlierCount());
  ASSERT_EQ(0.0f, detector.GetOrigin()[0]);
  for (int axis = 0; axis < 3; ++axis) { ASSERT_EQ(1.0f, detector.GetBounds(axis)[0]); ASSERT_EQ(-1.0f, detector.GetBounds(axis)[1]); }
}

TEST(Detector, Detect)   // Filter + 1.5-sigma removal + mean of |x| (detector.cu:120-188)
{
  // a 20x20x20 lattice of 1 cm pitch centred at (0.3, -0.2, 1.0) plus far points
  std::vector<Vector3f> host;
  for (int i = 0; i < 20; ++i)
    for (int j = 0; j < 20; ++j)
      for (int k = 0; k < 20; ++k)
        host.push_back(Vector3f(0.3f + 0.01f * (i - 9.5f), -0.2f + 0.01f * (j - 9.5f), 1.0f + 0.01f * (k - 9.5f)));
  for (int i = 0; i < 50; ++i) host.push_back(Vector3f(4.0f + i, 0, 0));   // outside the 2 m radius
  Buffer<Vector3f> points(host.size());
  points.CopyFromHost(host.data());

  Detector detector;
  const Vector3f position = detector.Detect(points);
  ASSERT_EQ(8000, detector.GetState().filtered_count);

  // host replay in double
  double centre[3] = {0, 0, 0};
  for (int i = 0; i < 8000; ++i) for (int a = 0; a < 3; ++a) centre[a] += std::fabs((double)host[i][a]) / 8000;
  double sq = 0;
  std::vector<double> dist(8000);
  for (int i = 0; i < 8000; ++i)
  {
    double d2 = 0;
    for (int a = 0; a < 3; ++a) d2 += ((double)host[i][a] - centre[a]) * ((double)host[i][a] - centre[a]);
    dist[i] = std::sqrt(d2);
    sq += d2;
  }
  const double limit = 1.5 * std::sqrt(sq / 8000);
  ASSERT_NEAR(limit, detector.GetState().limit, 1e-5);
  int survivors = 0;
  double mean[3] = {0, 0, 0};
  for (int i = 0; i < 8000; ++i)
  {
    ASSERT_TRUE(std::fabs(dist[i] - limit) > 1e-5);       // the lattice keeps clear of the threshold
    if (dist[i] <= limit) { ++survivors; for (int a = 0; a < 3; ++a) mean[a] += std::fabs((double)host[i][a]); }
  }
  ASSERT_EQ(survivors, detector.GetState().inlier_count);
  ASSERT_EQ(size_t(survivors), detector.GetInliers().GetSize());
  for (int a = 0; a < 3; ++a) ASSERT_NEAR(mean[a] / survivors, position[a], 1e-5);
  ASSERT_TRUE(position[1] > 0);                           // Sasum: the cloud sits at y = -0.2

  detector.SetMinInlierCount(survivors + 1);
  const Vector3f missing = detector.Detect(points);
  ASSERT_TRUE(std::isnan(missing[0]) && std::isnan(missing[1]) && std::isnan(missing[2]));
}

// ---- mesh extraction, export, sequences (upstream's extractor_test.cpp,
// exporter_test.cpp and mesh_test.cpp are empty) ---------------------------------------------

#include <fstream>
#include <sstream>
#include <sys/stat.h>

static std::shared_ptr<Volume> FusedPlane(Frame& frame)
{
  const int w = 160, h = 120;
  frame.depth_projection.SetFocalLength(136, 136);
  frame.depth_projection.SetCenterPoint(80, 60);
  frame.color_projection = frame.depth_projection;
  frame.depth_image = MakeDepth(w, h, [](int, int) { return 1.5f; });
  frame.color_image = MakeColor(w, h, [](int x, int y) { return Vector3f(0.1f + 0.005f * x, 0.2f + 0.006f * y, 0.3f); });
  auto volume = std::make_shared<Volume>(8192, 2048);
  volume->SetVoxelLength(0.008f);
  for (int i = 0; i < 6; ++i) volume->SetView(frame);
  DepthIntegrator integrator(volume);
  for (int i = 0; i < 3; ++i) integrator.Integrate(frame);
  return volume;
}

// Upstream's SetView reads the depth image only (volume.cu:430-437), so the normals or colours of a
// frame may legally change between SetView and Integrate. The light preparation that rides in
// SetView's request pass (vk_light_prep) must then not be used: it is tied to the images' content
// stamps (image.h), not to their addresses.
TEST(LightIntegrator, PreparationFollowsContentNotPointers)
{
  const int w = 160, h = 120;
  Light light;
  light.SetIntensity(2.0f);
  light.SetPosition(0.025f, 0.08f, 0.0f);
  std::vector<Voxel> results[2];
  for (int variant = 0; variant < 2; ++variant)
  {
    Frame frame;
    frame.depth_projection.SetFocalLength(136, 136);
    frame.depth_projection.SetCenterPoint(80, 60);
    frame.color_projection = frame.depth_projection;
    frame.depth_image = MakeDepth(w, h, [](int x, int) { return 1.5f + 0.001f * x; });
    frame.color_image = MakeColor(w, h, [](int x, int y) { return Vector3f(0.2f + 0.003f * x, 0.3f + 0.004f * y, 0.4f); });
    frame.ComputeNormals();
    auto volume = std::make_shared<Volume>(8192, 2048);
    volume->SetVoxelLength(0.008f);
    LightIntegrator integrator(volume);
    integrator.SetLight(light);
    for (int i = 0; i < 4; ++i) volume->SetView(frame);
    integrator.Integrate(frame);                       // registers the integrator's buffers with the volume
    // what the second integration has to see: other normals and other colours in the SAME buffers
    const std::vector<Vector3f> flipped(size_t(w) * h, Vector3f(0.0f, 0.6f, -0.8f));
    const std::vector<Vector3f> recolored(size_t(w) * h, Vector3f(0.5f, 0.25f, 0.75f));
    if (variant == 0)
    {
      volume->SetView(frame);                          // prepares mask + records from the OLD normals and colours
      frame.normal_image->CopyFromHost(flipped.data());
      frame.color_image->CopyFromHost(recolored.data());
    }
    else
    {
      frame.normal_image->CopyFromHost(flipped.data());
      frame.color_image->CopyFromHost(recolored.data());
      volume->SetView(frame);
    }
    integrator.Integrate(frame);
    results[variant] = Download(volume->GetVoxels());
  }
  ASSERT_EQ(results[0].size(), results[1].size());
  ASSERT_TRUE(std::memcmp(results[0].data(), results[1].data(), results[0].size() * sizeof(Voxel)) == 0);
  size_t coloured = 0;
  for (const Voxel& voxel : results[0]) coloured += voxel.color_weight > 1;
  ASSERT_TRUE(coloured > 1000);
}

// Volume::ComputeNormalsAndSetView (not upstream) = frame.ComputeNormals(); volume.SetView(frame, rounds):
// with a LightIntegrator attached the normals are computed inside SetView's request pass. Same normal
// image, same voxels.
TEST(Volume, ComputeNormalsAndSetViewEqualsTheTwoCalls)
{
  const int w = 160, h = 120;
  Light light;
  light.SetIntensity(2.0f);
  light.SetPosition(0.025f, 0.08f, 0.0f);
  std::vector<Voxel> voxels[2];
  std::vector<Vector3f> normals[2];
  for (int variant = 0; variant < 2; ++variant)
  {
    Frame frame;
    frame.depth_projection.SetFocalLength(136, 136);
    frame.depth_projection.SetCenterPoint(80, 60);
    frame.color_projection = frame.depth_projection;
    frame.depth_image = MakeDepth(w, h, [](int x, int y) { return 1.5f + 0.001f * x + 0.0007f * y; });
    frame.color_image = MakeColor(w, h, [](int x, int y) { return Vector3f(0.2f + 0.003f * x, 0.3f + 0.004f * y, 0.4f); });
    auto volume = std::make_shared<Volume>(8192, 2048);
    volume->SetVoxelLength(0.008f);
    LightIntegrator integrator(volume);
    integrator.SetLight(light);
    for (int i = 0; i < 3; ++i)     // (the first Integrate registers the integrator's buffers with the volume)
    {
      if (variant == 0)
      {
        frame.ComputeNormals();
        volume->SetView(frame, 3);
      }
      else volume->ComputeNormalsAndSetView(frame, 3);
      integrator.Integrate(frame);
    }
    voxels[variant] = Download(volume->GetVoxels());
    normals[variant].resize(size_t(w) * h);
    frame.normal_image->CopyToHost(normals[variant].data());
  }
  ASSERT_TRUE(std::memcmp(normals[0].data(), normals[1].data(), normals[0].size() * sizeof(Vector3f)) == 0);
  ASSERT_EQ(voxels[0].size(), voxels[1].size());
  ASSERT_TRUE(std::memcmp(voxels[0].data(), voxels[1].data(), voxels[0].size() * sizeof(Voxel)) == 0);
  size_t coloured = 0;
  for (const Voxel& voxel : voxels[0]) coloured += voxel.color_weight > 1;
  ASSERT_TRUE(coloured > 1000);
}

// a pool that runs dry must not go by in a counter. A 64 + 16 block pool in front of a frame that asks for
// thousands: requests are dropped (counted), the free-slot pointer falls below -1 as upstream's does (src/volume.cu:352-356),
// and the number of blocks in use is the capacity — not capacity - 1 - pointer, which overshoots by the dropped requests.
TEST(Volume, PoolExhaustionIsCountedAndTheAllocatedCountStaysInsideThePool)
{
  const int w = 160, h = 120;
  Frame frame;
  frame.depth_projection.SetFocalLength(136, 136);
  frame.depth_projection.SetCenterPoint(80, 60);
  frame.depth_image = MakeDepth(w, h, [](int x, int y) { return 1.5f + 0.001f * x + 0.0007f * y; });
  auto volume = std::make_shared<Volume>(64, 16);
  volume->SetVoxelLength(0.008f);
  volume->SetView(frame, 3);
  int32_t counters[VK_CTR_PUBLIC];
  volume->GetCounters(counters);           // (debug builds: says "memory exhausted" on stderr, once)
  ASSERT_TRUE(counters[VK_CTR_DROPPED] > 10);      // (the rounds stop after a round that drops: counted once, not three times)
  ASSERT_TRUE(counters[VK_CTR_VOXEL_PTR] < -1);
  ASSERT_EQ(80, volume->GetAllocatedBlockCount());
  // and the volume stays usable: integrate and raycast what did fit
  DepthIntegrator integrator(volume);
  integrator.Integrate(frame);
  Tracer tracer(volume);
  Frame traced;
  traced.depth_projection = frame.depth_projection;
  traced.depth_image = MakeDepth(w, h, [](int, int) { return 0.0f; });
  tracer.Trace(traced);
  Device::Synchronize();
  volume->SetView(frame, 3);
  volume->GetCounters(counters);
  ASSERT_EQ(80, volume->GetAllocatedBlockCount());
}

// Tracer::Trace(keyframe, next) — next_needs_normals defaults to false — followed by
// Volume::ComputeNormalsAndSetView(next). The announce left the normal image as it was (vk_requests_ahead.normals_made = 0):
// the normals are still due, and the preparation that rode with the pass was made from the OLD image. Same voxels and the
// same normal image as the sequence without an announce.
TEST(Volume, ComputeNormalsAndSetViewAfterAnAnnounceWithoutNormals)
{
  const int w = 160, h = 120;
  Light light;
  light.SetIntensity(2.0f);
  light.SetPosition(0.025f, 0.08f, 0.0f);
  std::vector<Voxel> voxels[3];
  std::vector<Vector3f> normals[3];
  for (int variant = 0; variant < 3; ++variant)       // 0: no announce; 1: announced without normals; 2: with
  {
    Frame frames[2];
    for (int i = 0; i < 2; ++i)
    {
      Frame& f = frames[i];
      f.depth_projection.SetFocalLength(136, 136);
      f.depth_projection.SetCenterPoint(80, 60);
      f.color_projection = f.depth_projection;
      f.depth_image = MakeDepth(w, h, [](int x, int y) { return 1.5f + 0.001f * x + 0.0007f * y; });
      f.color_image = MakeColor(w, h, [](int x, int y) { return Vector3f(0.2f + 0.003f * x, 0.3f + 0.004f * y, 0.4f); });
      f.depth_to_world_transform = Transform::Translate(0.004f * i, 0.0f, 0.0f);
    }
    // the second frame arrives with a normal image that holds something else (a recycled frame object)
    frames[1].normal_image = MakeColor(w, h, [](int, int) { return Vector3f(0.0f, 0.6f, -0.8f); });
    auto volume = std::make_shared<Volume>(8192, 2048);
    volume->SetVoxelLength(0.008f);
    LightIntegrator integrator(volume);
    integrator.SetLight(light);
    Tracer tracer(volume);
    volume->ComputeNormalsAndSetView(frames[0], 3);
    integrator.Integrate(frames[0]);
    Frame keyframe;
    keyframe.depth_projection = frames[0].depth_projection;
    keyframe.color_projection = frames[0].color_projection;
    keyframe.depth_to_world_transform = frames[0].depth_to_world_transform;
    keyframe.depth_image = MakeDepth(w, h, [](int, int) { return 0.0f; });
    if (variant == 0) tracer.Trace(keyframe);
    else
    {
      tracer.Trace(keyframe, frames[1], variant == 2);
      ASSERT_EQ(1, volume->GetRequestsAhead()->valid);
      ASSERT_EQ(variant == 2 ? 1 : 0, volume->GetRequestsAhead()->normals_made);
    }
    volume->ComputeNormalsAndSetView(frames[1], 3);
    ASSERT_EQ(0, volume->GetRequestsAhead()->valid);
    integrator.Integrate(frames[1]);
    voxels[variant] = Download(volume->GetVoxels());
    normals[variant].resize(size_t(w) * h);
    frames[1].normal_image->CopyToHost(normals[variant].data());
  }
  for (int variant = 1; variant < 3; ++variant)
  {
    ASSERT_TRUE(std::memcmp(normals[0].data(), normals[variant].data(), normals[0].size() * sizeof(Vector3f)) == 0);
    ASSERT_EQ(voxels[0].size(), voxels[variant].size());
    ASSERT_TRUE(std::memcmp(voxels[0].data(), voxels[variant].data(), voxels[0].size() * sizeof(Voxel)) == 0);
  }
  size_t coloured = 0;
  for (const Voxel& voxel : voxels[0]) coloured += voxel.color_weight > 1;
  ASSERT_TRUE(coloured > 1000);
}

TEST(Extractor, Extract)   // extractor.h:116-134; faces are what upstream leaves unwritten (extractor.cu:392-430)
{
  Frame frame;
  auto volume = FusedPlane(frame);
  Extractor extractor(volume);
  extractor.SetAllAllocated(true);
  Mesh mesh;
  extractor.Extract(mesh);
  ASSERT_TRUE(mesh.points.size() > 5000 && mesh.faces.size() > 10000);
  ASSERT_EQ(0, extractor.GetSkippedCubes());
  for (const Vector3f& p : mesh.points) ASSERT_NEAR(1.5f, p[2], 0.3f * 0.008f);     // on the plane
  int towards_camera = 0;
  for (const Vector3i& f : mesh.faces)
  {
    for (int k = 0; k < 3; ++k) ASSERT_TRUE(f[k] >= 0 && f[k] < (int)mesh.points.size());
    const Vector3f a = mesh.points[f[1]] - mesh.points[f[0]], b = mesh.points[f[2]] - mesh.points[f[0]];
    if (a[0] * b[1] - a[1] * b[0] < 0) ++towards_camera;        // normal z < 0: towards free space
  }
  ASSERT_TRUE(towards_camera > 0.999 * mesh.faces.size());

  // upstream's block source (the visible list) and vertex rule (edge midpoints)
  extractor.SetAllAllocated(false);
  extractor.SetInterpolate(false);
  Mesh visible;
  extractor.Extract(visible);
  ASSERT_TRUE(!visible.faces.empty() && visible.faces.size() <= mesh.faces.size());
  for (const Vector3f& p : visible.points) ASSERT_NEAR(1.5f, p[2], 0.5f * 0.008f + 1e-6f);
}

TEST(Exporter, Export)   // exporter.cpp:19-71, byte for byte
{
  Mesh mesh;
  mesh.points = { Vector3f(0.1f, -0.25f, 0.35f), Vector3f(1.5f, 2.0f, 0.85f), Vector3f(1e-7f, 123456.789f, 1.35f), Vector3f(0, 0, 0.6f) };
  mesh.faces = { Vector3i(0, 1, 2), Vector3i(2, 1, 3) };
  const std::string file = "/tmp/vulcan_host_test.ply";
  Exporter exporter(file);
  ASSERT_TRUE(exporter.GetFile() == file);
  exporter.Export(mesh);
  std::ifstream in(file);
  std::stringstream text;
  text << in.rdbuf();
  const std::string want =
      "ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
      "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 2\n"
      "property list uchar int vertex_indices\nend_header\n"
      "0.1 -0.25 0.35 0 0 0\n1.5 2 0.85 127 127 127\n1e-07 123457 1.35 255 255 255\n0 0 0.6 63 63 63\n"
      "3 0 1 2\n3 2 1 3\n";
  ASSERT_TRUE(text.str() == want);
}

TEST(Sequence, RoundTrip)   // image.h:100-133,228-253 Load / Save on PGM / PPM; sequence.h
{
  const int w = 64, h = 48;
  const std::string dir = "/tmp/vulcan_host_test_sequence";
  mkdir(dir.c_str(), 0755);
  Projection k;
  k.SetFocalLength(54.4162f, 54.43847f);
  k.SetCenterPoint(31.12701f, 23.47798f);
  std::vector<Transform> poses;
  {
    SequenceWriter writer(dir, w, h, k, k, 0.0002f);
    for (int i = 0; i < 3; ++i)
    {
      Frame frame;
      frame.depth_image = MakeDepth(w, h, [i](int x, int y) { return 0.5f + 0.01f * x + 0.02f * y + 0.1f * i; });
      frame.color_image = MakeColor(w, h, [](int x, int y) { return Vector3f(x / 64.0f, y / 48.0f, 0.5f); });
      frame.depth_to_world_transform = Transform::Translate(0.1f * i, -0.2f, 0.3f) * Transform::Rotate(0.9998719f, 0.0085884f, -0.0104268f, 0.0085884f);
      poses.push_back(frame.depth_to_world_transform);
      writer.Append(frame);
    }
    ASSERT_EQ(3, writer.GetFrameCount());
  }
  SequenceReader reader(dir);
  ASSERT_EQ(3, reader.GetFrameCount()); ASSERT_EQ(w, reader.GetWidth()); ASSERT_EQ(h, reader.GetHeight());
  for (int i = 0; i < 3; ++i)
  {
    Frame frame;
    reader.Read(i, frame);
    ASSERT_EQ(w, frame.depth_image->GetWidth()); ASSERT_EQ(h, frame.color_image->GetHeight());
    const std::vector<float> d = Download(*frame.depth_image);
    const std::vector<Vector3f> c = Download(*frame.color_image);
    for (int y = 0; y < h; y += 5)
      for (int x = 0; x < w; x += 7)
      {
        ASSERT_NEAR(0.5f + 0.01f * x + 0.02f * y + 0.1f * i, d[y * w + x], 0.0001f + 1e-6f);     // half a depth unit
        ASSERT_NEAR(x / 64.0f, c[y * w + x][0], 0.5f / 255 + 1e-6f);
      }
    ASSERT_FLOAT_EQ(54.4162f, frame.depth_projection.GetFocalLength()[0]);
    for (int r = 0; r < 4; ++r) for (int cc = 0; cc < 4; ++cc)
    {
      ASSERT_EQ(poses[i].GetMatrix()(r, cc), frame.depth_to_world_transform.GetMatrix()(r, cc));
      ASSERT_EQ(poses[i].GetInverseMatrix()(r, cc), frame.depth_to_world_transform.GetInverseMatrix()(r, cc));
    }
  }
  // Save: saturate(round-half-even(v * alpha + beta)) as cv::Mat::convertTo
  Image ramp(6, 1);
  const float values[6] = {-3.0f, 0.4999f, 0.5f, 1.5f, 2.5f, 300.0f};
  ramp.CopyFromHost(values);
  ramp.Save(dir + "/ramp.pgm", 8);
  Image back;
  back.Load(dir + "/ramp.pgm");
  const std::vector<float> r = Download(back);
  const float want[6] = {0, 0, 0, 2, 2, 255};
  for (int i = 0; i < 6; ++i) ASSERT_EQ(want[i], r[i]);
}

// ---- Tracer::Trace(frame, next_frame): no upstream counterpart; must equal Trace(frame) ... SetView(next_frame) ----

TEST(Tracer, TraceAnnouncingTheNextFrameEqualsTheTwoCalls)
{
  const int w = 160, h = 120, frames = 4;
  Light light;
  light.SetIntensity(2.0f);
  light.SetPosition(0.025f, 0.08f, 0.0f);
  for (int photometric = 0; photometric < 2; ++photometric)
  {
    std::vector<Voxel> voxels[2];
    std::vector<HashEntry> entries[2];
    std::vector<float> depths[2];
    std::vector<Vector3f> normals[2];
    for (int variant = 0; variant < 2; ++variant)
    {
      Frame frame;
      frame.depth_projection.SetFocalLength(136, 136);
      frame.depth_projection.SetCenterPoint(80, 60);
      frame.color_projection = frame.depth_projection;
      frame.depth_image = MakeDepth(w, h, [](int x, int y) { return 1.5f + 0.001f * x + 0.0007f * y; });
      frame.color_image = MakeColor(w, h, [](int x, int y) { return Vector3f(0.2f + 0.003f * x, 0.3f + 0.004f * y, 0.4f); });
      auto pose = [](int i) { return Transform::Translate(0.01f * i, -0.004f * i, 0.002f * i); };
      auto volume = std::make_shared<Volume>(8192, 2048);
      volume->SetVoxelLength(0.008f);
      DepthIntegrator depth_integrator(volume);
      LightIntegrator light_integrator(volume);
      light_integrator.SetLight(light);
      Tracer tracer(volume);
      Frame keyframe;
      keyframe.depth_projection = keyframe.color_projection = frame.depth_projection;
      keyframe.depth_image = std::make_shared<Image>(w, h);
      bool announced = false;
      for (int i = 0; i < frames; ++i)
      {
        frame.depth_to_world_transform = pose(i);
        if (photometric && !announced) frame.ComputeNormals();
        volume->SetView(frame, 3);
        ASSERT_EQ(0, volume->GetRequestsAhead()->valid);
        if (photometric) light_integrator.Integrate(frame); else depth_integrator.Integrate(frame);
        keyframe.depth_to_world_transform = frame.depth_to_world_transform;
        // (the first Integrate registers the light integrator's buffers with the volume: frame 0 announces like the rest)
        if (variant == 1 && i + 1 < frames)
        {
          Frame next = frame;
          next.depth_to_world_transform = pose(i + 1);
          tracer.Trace(keyframe, next, photometric != 0);
          ASSERT_EQ(1, volume->GetRequestsAhead()->valid);
          frame.normal_image = next.normal_image;
          announced = true;
          if (i == 1)
          {
            // any other frame is refused while the announced one is pending, and nothing is lost by the attempt
            Frame other = next;
            other.depth_to_world_transform = pose(7);
            bool thrown = false;
            try { volume->SetView(other, 3); } catch (const Exception&) { thrown = true; }
            ASSERT_TRUE(thrown);
            ASSERT_EQ(1, volume->GetRequestsAhead()->valid);
          }
        }
        else
        {
          tracer.Trace(keyframe);
          announced = false;
        }
      }
      voxels[variant] = Download(volume->GetVoxels());
      entries[variant] = Download(volume->GetHashEntries());
      depths[variant] = Download(*keyframe.depth_image);
      normals[variant].resize(size_t(w) * h);
      keyframe.normal_image->CopyToHost(normals[variant].data());
    }
    ASSERT_EQ(voxels[0].size(), voxels[1].size());
    ASSERT_TRUE(std::memcmp(voxels[0].data(), voxels[1].data(), voxels[0].size() * sizeof(Voxel)) == 0);
    ASSERT_TRUE(std::memcmp(entries[0].data(), entries[1].data(), entries[0].size() * sizeof(HashEntry)) == 0);
    ASSERT_TRUE(std::memcmp(depths[0].data(), depths[1].data(), depths[0].size() * sizeof(float)) == 0);
    ASSERT_TRUE(std::memcmp(normals[0].data(), normals[1].data(), normals[0].size() * sizeof(Vector3f)) == 0);
    size_t seen = 0;
    for (const Voxel& voxel : voxels[0]) seen += (photometric ? voxel.color_weight : voxel.distance_weight) > 1;
    ASSERT_TRUE(seen > 1000);
  }
}

// Tracer::TraceWithoutNormals + PyramidTracker<DepthTracker>::ComputeNormalsAndTrack(frame, true): the raycast's normal image
// made by the launch that builds the next Track's pyramid (round 5) — the same key-frame normals, the same tracked pose, bit
// for bit, as Tracer::Trace (tracer.cpp:41-47) followed by frame.ComputeNormals() and Track (vulcan.cu:297-311).
TEST(Tracer, TraceWithoutNormalsLeavesThemToTheNextTrack)
{
  const int w = 160, h = 120;
  std::vector<Vector3f> key_normals[2];
  Transform tracked[2];
  for (int variant = 0; variant < 2; ++variant)
  {
    Frame frame;
    frame.depth_projection.SetFocalLength(136, 136);
    frame.depth_projection.SetCenterPoint(80, 60);
    frame.color_projection = frame.depth_projection;
    frame.depth_image = MakeDepth(w, h, [&](int x, int y) { return float(1.4 + 0.08 * cos(5.0 * x / w) * sin(4.0 * y / h + 0.3)); });
    frame.color_image = MakeColor(w, h, [](int x, int y) { return Vector3f(0.2f + 0.003f * x, 0.3f + 0.004f * y, 0.4f); });
    auto volume = std::make_shared<Volume>(8192, 2048);
    volume->SetVoxelLength(0.008f);
    DepthIntegrator integrator(volume);
    Tracer tracer(volume);
    auto keyframe = std::make_shared<Frame>();
    keyframe->depth_projection = keyframe->color_projection = frame.depth_projection;
    keyframe->depth_image = std::make_shared<Image>(w, h);
    FuseToFixedPoint(volume, frame);
    integrator.Integrate(frame);
    if (variant == 0) tracer.Trace(*keyframe); else tracer.TraceWithoutNormals(*keyframe);
    Frame next = frame;
    next.normal_image.reset();
    next.depth_to_world_transform = Transform::Translate(0.002f, -0.001f, 0.0015f) * Transform::Rotate(0.999995f, 0.002f, -0.0015f, 0.001f);
    PyramidTracker<DepthTracker> tracker;
    tracker.SetKeyframe(keyframe);
    tracker.ComputeNormalsAndTrack(next, variant == 1);
    tracked[variant] = next.depth_to_world_transform;
    key_normals[variant].resize(size_t(w) * h);
    keyframe->normal_image->CopyToHost(key_normals[variant].data());
  }
  ASSERT_TRUE(std::memcmp(key_normals[0].data(), key_normals[1].data(), key_normals[0].size() * sizeof(Vector3f)) == 0);
  const Matrix4f A = tracked[0].GetMatrix(), B = tracked[1].GetMatrix();
  ASSERT_TRUE(std::memcmp(&A, &B, sizeof(A)) == 0);
  size_t with_normal = 0;
  for (const Vector3f& n : key_normals[1]) with_normal += (n[0] != 0 || n[1] != 0 || n[2] != 0) ? 1 : 0;
  ASSERT_TRUE(with_normal > 10000);
}

// Round 6: frame.ComputeNormals(); tracker.Track(frame); volume.SetView(frame, 3) as ONE call whose SetView is enqueued behind
// the Track at the pose the loop leaves on the device, before the host waits for it (Volume::SetViewAtDevicePose). The same
// pose, the same table, visible list and voxels as the three calls — with a LightIntegrator's preparation riding in both.
TEST(PyramidTracker, ComputeNormalsTrackAndSetViewEqualsTheThreeCalls)
{
  const int w = 160, h = 120;
  Light light;
  light.SetIntensity(2.0f);
  light.SetPosition(0.025f, 0.08f, 0.0f);
  std::vector<Voxel> voxels[2];
  std::vector<HashEntry> entries[2];
  Transform tracked[2];
  int visible[2] = {0, 0};
  for (int variant = 0; variant < 2; ++variant)
  {
    Frame frame;
    frame.depth_projection.SetFocalLength(136, 136);
    frame.depth_projection.SetCenterPoint(80, 60);
    frame.color_projection = frame.depth_projection;
    frame.depth_image = MakeDepth(w, h, [&](int x, int y) { return float(1.4 + 0.08 * cos(5.0 * x / w) * sin(4.0 * y / h + 0.3)); });
    frame.color_image = MakeColor(w, h, [](int x, int y) { return Vector3f(0.2f + 0.003f * x, 0.3f + 0.004f * y, 0.4f); });
    auto volume = std::make_shared<Volume>(8192, 2048);
    volume->SetVoxelLength(0.008f);
    LightIntegrator integrator(volume);
    integrator.SetLight(light);
    Tracer tracer(volume);
    auto keyframe = std::make_shared<Frame>();
    keyframe->depth_projection = keyframe->color_projection = frame.depth_projection;
    keyframe->depth_image = std::make_shared<Image>(w, h);
    volume->ComputeNormalsAndSetView(frame, 3);
    integrator.Integrate(frame);
    tracer.TraceWithoutNormals(*keyframe);
    Frame next = frame;
    next.normal_image = std::make_shared<ColorImage>(w, h);
    next.depth_to_world_transform = Transform::Translate(0.002f, -0.001f, 0.0015f) * Transform::Rotate(0.999995f, 0.002f, -0.0015f, 0.001f);
    PyramidTracker<DepthTracker> tracker;
    tracker.SetKeyframe(keyframe);
    vk_test_hooks hooks;
    VK_ASSERT(vk_test_hooks_get(&hooks));
    if (variant == 0)
    {
      if (hooks.force_loop_abort == 1)
      {
        // (host_tests Track --force-loop-abort: the one-launch loops abort and the Track is repeated stage by stage. The
        // combined call's SetView has then run at the START pose the aborted loop left on the device — the documented
        // state: the volume has seen SetView(frame at the start pose) before the frame's own SetView)
        next.ComputeNormals();
        volume->SetView(next, 3);
      }
      tracker.ComputeNormalsAndTrack(next, true);
      volume->SetView(next, 3);
    }
    else tracker.ComputeNormalsTrackAndSetView(next, *volume, 3, true);
    tracked[variant] = next.depth_to_world_transform;
    integrator.Integrate(next);
    voxels[variant] = Download(volume->GetVoxels());
    entries[variant] = Download(volume->GetHashEntries());
    int32_t counters[VK_CTR_PUBLIC];
    volume->GetCounters(counters);
    visible[variant] = counters[VK_CTR_VISIBLE];
  }
  const Matrix4f A = tracked[0].GetMatrix(), B = tracked[1].GetMatrix();
  ASSERT_TRUE(std::memcmp(&A, &B, sizeof(A)) == 0);
  ASSERT_EQ(visible[0], visible[1]);
  ASSERT_TRUE(visible[0] > 100);
  ASSERT_EQ(entries[0].size(), entries[1].size());
  ASSERT_TRUE(std::memcmp(entries[0].data(), entries[1].data(), entries[0].size() * sizeof(HashEntry)) == 0);
  ASSERT_EQ(voxels[0].size(), voxels[1].size());
  ASSERT_TRUE(std::memcmp(voxels[0].data(), voxels[1].data(), voxels[0].size() * sizeof(Voxel)) == 0);
}

// ---- FrameUploader (upload.h): no upstream test — upstream uploads with a blocking copy (image.h:100-123) ----

TEST(FrameUploader, DeliversEveryFrameInOrderThroughTwoSlots)
{
  const int w = 64, h = 48, frames = 7;
  FrameUploader uploader(w, h, true);
  Frame frame;
  auto fill = [&](int i) {
    float* d = uploader.StagingDepth();
    Vector3f* c = uploader.StagingColor();
    for (int p = 0; p < w * h; ++p)
    {
      d[p] = 1.0f + 0.001f * i + 1e-6f * p;
      c[p] = Vector3f(0.1f * i, 0.5f + 1e-5f * p, 0.25f);
    }
  };
  fill(0);
  uploader.Submit();
  for (int i = 0; i < frames; ++i)
  {
    if (i + 1 < frames) { fill(i + 1); uploader.Submit(); }           // frame i + 1 crosses while frame i is used
    uploader.Acquire(frame);
    ASSERT_EQ(w, frame.depth_image->GetWidth());
    // a reader on the compute stream: the normals of the uploaded depth image (any kernel would do)
    frame.depth_projection.SetFocalLength(60.0f, 60.0f);
    frame.depth_projection.SetCenterPoint(32.0f, 24.0f);
    frame.ComputeNormals();
    std::vector<float> depth = Download(*frame.depth_image);
    std::vector<Vector3f> color(size_t(w) * h);
    frame.color_image->CopyToHost(color.data());
    uploader.Release();
    for (int p = 0; p < w * h; p += 97)
    {
      ASSERT_EQ(1.0f + 0.001f * i + 1e-6f * p, depth[p]);
      ASSERT_EQ(0.1f * i, color[p][0]);
      ASSERT_EQ(0.5f + 1e-5f * p, color[p][1]);
    }
  }
  ASSERT_EQ(frames, uploader.GetSubmitted());
}

int main(int argc, char** argv)
{
  int count = 0;
  VK_ASSERT(vk_device_count(&count));
  if (count == 0) { std::printf("host_tests: no HIP device\n"); return 2; }
  // host_tests [filter] [--force-loop-abort]: the flag sets vk_test_hooks.force_loop_abort, under which every
  // one-launch Gauss-Newton loop ends with VK_TRACK_ABORTED and the trackers take the launch-per-stage path
  std::string filter;
  for (int i = 1; i < argc; ++i)
  {
    if (std::string(argv[i]) == "--force-loop-abort")
    {
      vk_test_hooks hooks;
      VK_ASSERT(vk_test_hooks_get(&hooks));
      hooks.force_loop_abort = 1;
      VK_ASSERT(vk_test_hooks_set(&hooks));
    }
    else filter = argv[i];
  }
  int failed = 0, ran = 0;
  for (const TestCase& t : Registry())
  {
    if (!filter.empty() && std::string(t.name).find(filter) == std::string::npos) continue;
    ++ran;
    try { t.body(); Device::Synchronize(); std::printf("[  OK  ] %s\n", t.name); }
    catch (const Failure& f) { ++failed; std::printf("[FAILED] %s\n         %s\n", t.name, f.text.c_str()); }
    catch (const std::exception& e) { ++failed; std::printf("[FAILED] %s\n         exception: %s\n", t.name, e.what()); }
    std::fflush(stdout);
  }
  std::printf("%d test(s), %d failed\n", ran, failed);
  return failed ? 1 : 0;
}
