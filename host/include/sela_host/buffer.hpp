// buffer.hpp -- PinnedBuffer<T>: a flat, contiguous, trivially-copyable-element array in page-locked host
// memory (sela_hip_host_alloc), the storage of the big byte arrays of the host classes (WAV data chunk,
// .sela frame stream).  Copies between such memory and the GPU are asynchronous, which is what lets
// sela::Encoder / sela::Decoder keep reading the file while earlier pieces are already on the device.
// Unlike std::vector it does not value-initialise on resize (a 32 MB memset costs more than encoding it),
// and it falls back to ordinary memory where no GPU is present (container code still runs on the CPU).
#pragma once

#include <cstddef>
#include <cstring>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "sela_hip.h"

namespace sela_host {

template <typename T>
class PinnedBuffer {
    static_assert(std::is_trivially_copyable<T>::value, "PinnedBuffer holds plain data");
    T* ptr = nullptr;
    size_t count = 0, cap = 0;

public:
    PinnedBuffer() {}
    explicit PinnedBuffer(size_t n) { resize(n); }
    PinnedBuffer(const std::vector<T>& v) { assign(v.data(), v.size()); }
    PinnedBuffer(const PinnedBuffer& o) { assign(o.ptr, o.count); }
    PinnedBuffer(PinnedBuffer&& o) noexcept : ptr(o.ptr), count(o.count), cap(o.cap) { o.ptr = nullptr, o.count = o.cap = 0; }
    PinnedBuffer& operator=(const PinnedBuffer& o)
    {
        if (this != &o)
            assign(o.ptr, o.count);
        return *this;
    }
    PinnedBuffer& operator=(PinnedBuffer&& o) noexcept
    {
        if (this != &o) {
            release();
            ptr = o.ptr, count = o.count, cap = o.cap;
            o.ptr = nullptr, o.count = o.cap = 0;
        }
        return *this;
    }
    ~PinnedBuffer() { release(); }

    T* data() { return ptr; }
    const T* data() const { return ptr; }
    size_t size() const { return count; }
    bool empty() const { return count == 0; }
    T& operator[](size_t i) { return ptr[i]; }
    const T& operator[](size_t i) const { return ptr[i]; }
    T* begin() { return ptr; }
    T* end() { return ptr + count; }
    const T* begin() const { return ptr; }
    const T* end() const { return ptr + count; }

    // grow or shrink; new elements are NOT initialised, old ones are kept
    void resize(size_t n)
    {
        if (n > cap) {
            T* fresh = static_cast<T*>(sela_hip_host_alloc(n * sizeof(T)));
            if (!fresh)
                throw std::bad_alloc();
            if (count)
                std::memcpy(fresh, ptr, count * sizeof(T));
            if (ptr)
                sela_hip_host_free(ptr);
            ptr = fresh;
            cap = n;
        }
        count = n;
    }
    void assign(const T* src, size_t n)
    {
        count = 0;
        resize(n);
        if (n)
            std::memcpy(ptr, src, n * sizeof(T));
    }
    void release()
    {
        if (ptr)
            sela_hip_host_free(ptr);
        ptr = nullptr;
        count = cap = 0;
    }
    bool operator==(const PinnedBuffer& o) const { return count == o.count && (count == 0 || std::memcmp(ptr, o.ptr, count * sizeof(T)) == 0); }
    bool operator==(const std::vector<T>& o) const { return count == o.size() && (count == 0 || std::memcmp(ptr, o.data(), count * sizeof(T)) == 0); }
};

} // namespace sela_host
