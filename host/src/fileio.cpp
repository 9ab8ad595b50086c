// fileio.cpp -- see fileio.hpp.
#include "sela_host/fileio.hpp"

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <deque>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "sela_host/data.hpp"

namespace sela_host {

void (*ioTrace)(const char*, long long, long long, size_t) = nullptr;
long long ioNow() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

namespace {

struct Traced { // reports its lifetime when the hook is set
    const char* what;
    size_t bytes;
    long long t0;
    Traced(const char* w, size_t b) : what(w), bytes(b), t0(ioTrace ? ioNow() : 0) {}
    ~Traced()
    {
        if (ioTrace)
            ioTrace(what, t0, ioNow(), bytes);
    }
};

[[noreturn]] void ioFailure(const std::string& what, const std::string& path)
{
    throw data::Exception(what + " " + path + ": " + std::strerror(errno));
}

std::atomic<unsigned> g_configured{ 0 };

} // namespace

PosixFile& PosixFile::operator=(PosixFile&& o) noexcept
{
    if (this != &o) {
        close();
        fd = o.fd;
        name = std::move(o.name);
        o.fd = -1;
    }
    return *this;
}

PosixFile PosixFile::openForRead(const std::string& path)
{
    PosixFile f;
    f.fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (f.fd < 0)
        throw data::Exception("cannot open " + path);
    f.name = path;
    return f;
}

PosixFile PosixFile::create(const std::string& path)
{
    PosixFile f;
    f.fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (f.fd < 0)
        throw data::Exception("cannot open " + path + " for writing");
    f.name = path;
    return f;
}

PosixFile PosixFile::openForWrite(const std::string& path)
{
    PosixFile f;
    f.fd = ::open(path.c_str(), O_WRONLY | O_CLOEXEC);
    if (f.fd < 0)
        throw data::Exception("cannot open " + path + " for writing");
    f.name = path;
    return f;
}

size_t PosixFile::size() const
{
    struct stat st;
    if (::fstat(fd, &st) != 0)
        ioFailure("cannot stat", name);
    return st.st_size > 0 ? (size_t)st.st_size : 0;
}

bool PosixFile::readAt(void* dst, size_t n, size_t offset) const
{
    char* p = static_cast<char*>(dst);
    while (n) {
        const ssize_t got = ::pread(fd, p, n, (off_t)offset);
        if (got < 0) {
            if (errno == EINTR)
                continue;
            ioFailure("reading", name);
        }
        if (got == 0)
            return false;
        p += got, offset += (size_t)got, n -= (size_t)got;
    }
    return true;
}

void PosixFile::writeAt(const void* src, size_t n, size_t offset) const
{
    const char* p = static_cast<const char*>(src);
    while (n) {
        const ssize_t put = ::pwrite(fd, p, n, (off_t)offset);
        if (put < 0) {
            if (errno == EINTR)
                continue;
            ioFailure("writing", name);
        }
        p += put, offset += (size_t)put, n -= (size_t)put;
    }
}

void PosixFile::allocate(size_t from, size_t n) const
{
    if (n == 0)
        return;
    int rc;
    do // (the Linux call, not posix_fallocate: where the file system cannot do it, that one writes zeros instead)
        rc = ::fallocate(fd, FALLOC_FL_KEEP_SIZE, (off_t)from, (off_t)n); // (KEEP_SIZE: a late call can never lengthen a file that has been cut to size)
    while (rc != 0 && errno == EINTR);
    if (rc != 0)
        ioFailure("allocating", name);
}

void PosixFile::truncate(size_t n) const
{
    if (::ftruncate(fd, (off_t)n) != 0)
        ioFailure("truncating", name);
}

void PosixFile::close()
{
    if (fd >= 0)
        (void)::close(fd);
    fd = -1;
}

// ---- the pool ------------------------------------------------------------------------------------------------------
struct IoPool::Impl {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> queue;
    std::vector<std::thread> workers;
    bool stopping = false;
};

void IoPool::configure(unsigned n) { g_configured.store(n, std::memory_order_relaxed); }

IoPool::IoPool() : impl(new Impl)
{
    unsigned n = g_configured.load(std::memory_order_relaxed);
    if (n == 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        n = std::min(6u, std::max(2u, hw / 2));
    }
    else
        userSized = true;
    n = std::min(n, 256u);
    count.store(n, std::memory_order_relaxed);
    for (unsigned i = 0; i < n; i++)
        impl->workers.emplace_back([this] { run(); });
}

IoPool::~IoPool()
{
    {
        std::lock_guard<std::mutex> lock(impl->mu);
        impl->stopping = true;
    }
    impl->cv.notify_all();
    for (std::thread& t : impl->workers)
        t.join();
    delete impl;
}

void IoPool::grow(unsigned atLeast)
{
    if (userSized)
        return;
    atLeast = std::min(atLeast, 256u);
    std::lock_guard<std::mutex> lock(impl->mu);
    while (count.load(std::memory_order_relaxed) < atLeast) {
        impl->workers.emplace_back([this] { run(); });
        count.fetch_add(1, std::memory_order_relaxed);
    }
}

IoPool& IoPool::instance()
{
    static IoPool pool;
    return pool;
}

void IoPool::submit(std::function<void()> task, bool first)
{
    {
        std::lock_guard<std::mutex> lock(impl->mu);
        if (first)
            impl->queue.push_front(std::move(task));
        else
            impl->queue.push_back(std::move(task));
    }
    impl->cv.notify_one();
}

void IoPool::run()
{
    for (;;) {
        std::function<void()> task;
        {
            std::unique_lock<std::mutex> lock(impl->mu);
            impl->cv.wait(lock, [this] { return impl->stopping || !impl->queue.empty(); });
            if (impl->queue.empty())
                return; // stopping
            task = std::move(impl->queue.front());
            impl->queue.pop_front();
        }
        task(); // (IoGroup::run wraps every task: nothing escapes)
    }
}

// ---- groups --------------------------------------------------------------------------------------------------------
void IoGroup::run(std::function<void()> task)
{
    {
        std::lock_guard<std::mutex> lock(mu);
        pending++;
    }
    IoPool::instance().submit([this, task = std::move(task)] {
        std::string what;
        bool bad = false;
        try {
            task();
        } catch (const data::Exception& e) {
            bad = true, what = e.exceptionMessage;
        } catch (const std::exception& e) {
            bad = true, what = e.what();
        }
        std::lock_guard<std::mutex> lock(mu);
        if (bad && !failed)
            failed = true, error = what;
        if (--pending == 0)
            cv.notify_all();
    });
}

void IoGroup::waitNoThrow()
{
    std::unique_lock<std::mutex> lock(mu);
    cv.wait(lock, [this] { return pending == 0; });
}

void IoGroup::wait()
{
    waitNoThrow();
    std::lock_guard<std::mutex> lock(mu);
    if (failed) {
        failed = false;
        throw data::Exception(error);
    }
}

bool IoGroup::hasFailed()
{
    std::lock_guard<std::mutex> lock(mu);
    return failed;
}

// ---- read ahead ----------------------------------------------------------------------------------------------------
ReadAhead::ReadAhead(const PosixFile& f, void* dstBase, size_t fileOffset, size_t total, size_t pieceBytes, size_t subBytes)
    : file(f), dst(static_cast<uint8_t*>(dstBase)), fileOffset(fileOffset), total(total), piece(std::max<size_t>(pieceBytes, 1)),
      left((total + std::max<size_t>(pieceBytes, 1) - 1) / std::max<size_t>(pieceBytes, 1))
{
    subBytes = std::max<size_t>(subBytes, 4096);
    for (size_t p = 0; p < left.size(); p++) {
        const size_t begin = p * piece, end = std::min(total, begin + piece);
        left[p].store((uint32_t)((end - begin + subBytes - 1) / subBytes), std::memory_order_relaxed);
    }
    for (size_t p = 0; p < left.size(); p++) {
        const size_t begin = p * piece, end = std::min(total, begin + piece);
        for (size_t at = begin; at < end; at += subBytes) {
            const size_t n = std::min(subBytes, end - at);
            group.run([this, p, at, n] {
                bool ok = false;
                std::string what;
                try {
                    Traced t("pread", n);
                    ok = file.readAt(dst + at, n, this->fileOffset + at);
                } catch (const data::Exception& e) {
                    what = e.exceptionMessage;
                }
                {
                    std::lock_guard<std::mutex> lock(mu);
                    if (!what.empty() && error.empty())
                        error = what;
                    else if (!ok && what.empty())
                        shortFile = true;
                    left[p].fetch_sub(1, std::memory_order_release);
                }
                cv.notify_all();
            });
        }
    }
}

void ReadAhead::need(size_t upTo)
{
    Traced t("wait for reads", upTo);
    upTo = std::min(upTo, total);
    const size_t lastPiece = upTo ? (upTo - 1) / piece : 0;
    std::unique_lock<std::mutex> lock(mu);
    for (size_t p = 0; upTo && p <= lastPiece; p++)
        cv.wait(lock, [this, p] { return left[p].load(std::memory_order_acquire) == 0; });
    if (!error.empty())
        throw data::Exception(error);
    if (shortFile)
        throw data::Exception("file " + file.path() + " is shorter than its header says");
}

// ---- write behind --------------------------------------------------------------------------------------------------
WriteBehind::WriteBehind(const PosixFile& f, size_t fileOffset, size_t subBytes, size_t expectBytes)
    : file(f), fileOffset(fileOffset), subBytes(std::max<size_t>(subBytes, 4096)), expect(expectBytes)
{
    if (expect) { // the pages of a fresh file, in one go, while the data is still on its way
        std::lock_guard<std::mutex> lock(mu);
        active = true;
        IoPool::instance().submit([this] { strand(); }, true); // (ahead of the reads that are queued already: it has 1-2 ms of work that needs no data)
    } else
        allocated = true;
}

WriteBehind::~WriteBehind()
{
    std::unique_lock<std::mutex> lock(mu);
    cv.wait(lock, [this] { return !active; });
}

void WriteBehind::strand()
{
    std::unique_lock<std::mutex> lock(mu);
    for (;;) {
        std::string what;
        if (!allocated) {
            // the file's pages, in ONE call (4 MB at a time between the writes took twice as long per byte), before the
            // first byte is there: this runs while the input is being read and coded
            lock.unlock();
            try {
                Traced t("fallocate", expect);
                file.allocate(fileOffset, expect);
            } catch (const data::Exception&) {
                // (a file system without fallocate: the writes allocate as they go)
            }
            lock.lock();
            allocated = true;
            continue;
        }
        if (failed || written >= target)
            break;
        const size_t at = written, n = std::min(subBytes, target - written);
        const uint8_t* from = base;
        lock.unlock();
        try {
            Traced t("pwrite", n);
            file.writeAt(from + at, n, fileOffset + at);
        } catch (const data::Exception& e) {
            what = e.exceptionMessage;
        }
        lock.lock();
        if (!what.empty()) {
            failed = true;
            error = what;
            break;
        }
        written = at + n;
    }
    active = false;
    cv.notify_all();
}

void WriteBehind::drain(const void* from, size_t upTo)
{
    std::lock_guard<std::mutex> lock(mu);
    base = static_cast<const uint8_t*>(from);
    if (upTo > target)
        target = upTo;
    if (!active && written < target && !failed) {
        active = true;
        IoPool::instance().submit([this] { strand(); }, true);
    }
}

void WriteBehind::quiesce()
{
    std::unique_lock<std::mutex> lock(mu);
    cv.wait(lock, [this] { return !active; });
}

void WriteBehind::finish(const size_t* truncateTo)
{
    Traced t("wait for writes", 0);
    {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [this] { return !active; });
        if (failed)
            throw data::Exception(error);
    }
    if (truncateTo && expect) // (the allocation keeps the file's size: the length is set here, once)
        file.truncate(fileOffset + *truncateTo);
}

} // namespace sela_host
