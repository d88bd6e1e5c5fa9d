// image.h — single-channel float image and packed float3 image in device memory
// (ref: include/vulcan/image.h). Resize discards contents when the pixel count changes
// (image.h:85-97). Load / Save keep the reference's signatures (image.h:100-133,228-253)
// for the two formats that need no OpenCV — binary PGM (8 / 16 bit) and binary PPM —
// with cv::Mat::convertTo's arithmetic: Load = pixel * scale, Save = saturate(round-half-
// even(v * alpha + beta)).
#pragma once

#include <atomic>
#include <stdint.h>
#include <string>
#include <vulcan/device.h>
#include <vulcan/matrix.h>

namespace vulcan
{

namespace detail
{

// A process-wide sequence of stamps: an image takes a new one whenever its pixels may have
// changed, so two equal stamps mean "the same image, untouched in between" (vk_frame.content_id).
inline uint64_t NextContentStamp()
{
  static std::atomic<uint64_t> next(1);
  return next.fetch_add(1, std::memory_order_relaxed);
}

// shared storage logic of Image (1 float / pixel) and ColorImage (3 floats / pixel)
template <typename Pixel>
class ImageStorage
{
  public:

    ImageStorage() : size_(0, 0), data_(nullptr), capacity_(0), stamp_(NextContentStamp()) {}

    ImageStorage(int w, int h) : size_(0, 0), data_(nullptr), capacity_(0), stamp_(NextContentStamp()) { Resize(w, h); }

    ~ImageStorage() { vk_free(data_); }

    int GetWidth() const { return size_[0]; }

    int GetHeight() const { return size_[1]; }

    const Vector2i& GetSize() const { return size_; }

    int GetTotal() const { return size_[0] * size_[1]; }

    int GetBytes() const { return sizeof(Pixel) * GetTotal(); }

    const Pixel* GetData() const { return data_; }

    // a pointer that can be written through: the content counts as changed from here on (a
    // caller that keeps the pointer and writes later must call Touch() after writing)
    Pixel* GetData() { Touch(); return data_; }

    // not upstream: identity of the content, see NextContentStamp
    uint64_t GetContentStamp() const { return stamp_; }

    void Touch() { stamp_ = NextContentStamp(); }

    void Resize(int w, int h) { Resize(Vector2i(w, h)); }

    void Resize(const Vector2i& size)
    {
      VULCAN_DEBUG(size[0] >= 0 && size[1] >= 0);
      if (size[0] != size_[0] || size[1] != size_[1]) Touch();
      size_ = size;
      // upstream frees and allocates whenever the pixel count changes (image.h:85-97); a
      // pyramid tracker alternates between two sizes every frame, and on this runtime a free
      // is a device-wide synchronisation: the storage only ever grows
      if ((size_t)GetTotal() <= capacity_) return;
      VK_ASSERT(vk_free(data_));
      data_ = nullptr;
      capacity_ = 0;
      void* ptr = nullptr;
      VK_ASSERT(vk_malloc(&ptr, GetBytes()));
      data_ = static_cast<Pixel*>(ptr);
      capacity_ = (size_t)GetTotal();
    }

    void CopyFromHost(const Pixel* pixels)
    {
      VK_ASSERT(vk_memcpy_h2d(data_, pixels, GetBytes(), Device::GetStream()));
      Touch();
    }

    // not upstream (whose Load / CopyFromHost block): the copy is ENQUEUED on `copy_stream` from pinned host memory
    // (vk_malloc_host) and the call returns; the caller orders it against the kernels that read the image with events
    // (vk.h "the input side of a frame"; FrameUploader in upload.h does all of that)
    void CopyFromHostAsync(const Pixel* pinned_pixels, void* copy_stream)
    {
      VK_ASSERT(vk_memcpy_h2d_async(data_, pinned_pixels, GetBytes(), copy_stream));
      Touch();
    }

    void CopyToHost(Pixel* pixels) const
    {
      VK_ASSERT(vk_memcpy_d2h(pixels, data_, GetBytes(), Device::GetStream()));
    }

  private:

    ImageStorage(const ImageStorage&);

    ImageStorage& operator=(const ImageStorage&);

  protected:

    Vector2i size_;

    Pixel* data_;

    size_t capacity_;   // pixels allocated (>= GetTotal())

    uint64_t stamp_;    // GetContentStamp()
};

} // namespace detail

class Image : public detail::ImageStorage<float>
{
  public:

    Image() {}

    Image(int w, int h) : detail::ImageStorage<float>(w, h) {}

    // ref: image.cu:183-211 — 2x nearest or 2x2 box
    void Downsample(Image& image, bool nearest) const;

    // 3x3 smoothed central differences, zero padding (ref: image.cu:21-99,166-179)
    void GetGradients(Image& gx, Image& gy) const;

    // ref: image.h:100-111 — grey PGM (an RGB PPM is reduced to grey first, as RGB2GRAY does)
    void Load(const std::string& file, float scale = 1);

    // ref: image.cu:213-221 — bits = 8 (CV_8UC1) or 16 (CV_16UC1)
    void Save(const std::string& file, int bits = 8, float alpha = 1, float beta = 0) const;
};

class ColorImage : public detail::ImageStorage<Vector3f>
{
  public:

    ColorImage() {}

    ColorImage(int w, int h) : detail::ImageStorage<Vector3f>(w, h) {}

    // ref: image.cu:234-262
    void Downsample(ColorImage& image, bool nearest) const;

    // intensity = (r + g + b) / 3 (ref: image.cu:10-19,235-247)
    void ConvertTo(Image& image) const;

    // ref: image.h:228-240 — PPM in R, G, B order (a grey PGM is replicated to three channels)
    void Load(const std::string& file, float scale = 1);

    // ref: image.cu:264-273
    void Save(const std::string& file, int bits = 8, float alpha = 1, float beta = 0) const;
};

} // namespace vulcan
