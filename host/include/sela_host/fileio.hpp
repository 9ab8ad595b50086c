// fileio.hpp -- positioned file I/O on a small pool of threads, for the path-based verbs of codec.hpp.
//
// What it replaces in the reference: `ifstream::read` of the whole file before the first frame is coded and
// field-by-field `ofstream::write` after the last (src/file/wav_file.cpp:39-45, src/file/sela_file.cpp:105-137).
// At GPU speed the file system is the slow side: one thread moves a page-cache file at a few GB/s (a fresh file
// also pays a page allocation per 4 KB written), the device codes the same bytes ten times faster.  So a file is
// read with several pread()s in flight into page-locked memory while earlier pieces are already on the device, and
// finished ranges are written behind the feeding thread's back while later pieces are still being coded -- by one
// task per output file (writes to one file do not run side by side, see WriteBehind), over pages allocated up front.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

namespace sela_host {

// Measurement hook (host/sela_filebench ... trace): when set, every pool task and the feeding thread's waits report
// (what, start, end) in steady_clock nanoseconds, bytes.  Null in every other program.
extern void (*ioTrace)(const char* what, long long t0_ns, long long t1_ns, size_t bytes);
long long ioNow();

// A file descriptor with exact positioned reads and writes.  Throws data::Exception.
class PosixFile {
    int fd = -1;
    std::string name;

public:
    PosixFile() {}
    PosixFile(const PosixFile&) = delete;
    PosixFile& operator=(const PosixFile&) = delete;
    PosixFile(PosixFile&& o) noexcept : fd(o.fd), name(std::move(o.name)) { o.fd = -1; }
    PosixFile& operator=(PosixFile&& o) noexcept;
    ~PosixFile() { close(); }

    static PosixFile openForRead(const std::string& path);
    static PosixFile create(const std::string& path); // truncates
    static PosixFile openForWrite(const std::string& path); // an existing file, contents kept
    bool isOpen() const { return fd >= 0; }
    const std::string& path() const { return name; }
    size_t size() const;
    // false when the file ends before n bytes at `offset`
    bool readAt(void* dst, size_t n, size_t offset) const;
    void writeAt(const void* src, size_t n, size_t offset) const;
    void allocate(size_t from, size_t n) const; // the pages of [from, from + n), in one go (fallocate, FALLOC_FL_KEEP_SIZE: the file's LENGTH stays what the writes and truncate() make it)
    void truncate(size_t n) const;
    void close();
};

// The process-wide pool of I/O threads (started on first use).  Tasks run in the order they were submitted.
class IoPool {
public:
    static IoPool& instance();
    void submit(std::function<void()> task, bool first = false); // first: ahead of what is queued
    unsigned threads() const { return count.load(std::memory_order_relaxed); }
    // more threads (never fewer): the batch verbs ask for a few per GPU worker.  A thread count the user configured
    // (configure(n), --io-threads) is a decision, not a default: the pool then stays at n
    void grow(unsigned atLeast);
    // Threads the pool starts with (before its first use; later calls are ignored).  0 = the default:
    // min(6, hardware threads / 2), at least 2 (reads of one file stop scaling at 4-8 threads; writes to one file
    // do not scale at all, see WriteBehind).
    static void configure(unsigned n);

private:
    IoPool();
    ~IoPool();
    void run();
    struct Impl;
    Impl* impl;
    std::atomic<unsigned> count{ 0 };
    bool userSized = false;
};

// A set of tasks submitted to the pool whose completion is awaited together; remembers the first failure
// (a data::Exception's message) and rethrows it from wait().
class IoGroup {
    std::mutex mu;
    std::condition_variable cv;
    size_t pending = 0;
    std::string error;
    bool failed = false;

public:
    IoGroup() {}
    IoGroup(const IoGroup&) = delete;
    ~IoGroup() { waitNoThrow(); }
    void run(std::function<void()> task);
    void wait(); // throws data::Exception if a task failed
    void waitNoThrow();
    bool hasFailed();
};

// A byte range of a file being read into memory by pool tasks, consumed front to back: need(n) returns once
// bytes [0, n) have arrived (or throws if the file turned out shorter / a read failed).
class ReadAhead {
    const PosixFile& file;
    uint8_t* dst;
    size_t fileOffset, total, piece;
    std::vector<std::atomic<uint32_t>> left; // sub-reads still out per piece
    std::mutex mu;
    std::condition_variable cv;
    bool shortFile = false;
    std::string error;
    IoGroup group;

public:
    // Reads file[fileOffset, fileOffset + total) into dst in pieces of `pieceBytes`, each cut into sub-reads of
    // `subBytes`; every sub-read is a pool task, submitted now, in order.
    ReadAhead(const PosixFile& f, void* dst, size_t fileOffset, size_t total, size_t pieceBytes, size_t subBytes);
    void need(size_t upTo);
    void finish() { group.wait(); }
};

// Ranges of a memory buffer that have become final, written to a file behind the caller's back -- by ONE pool task at a
// time, in order: buffered writes to one file hold the inode's lock, so several pwrite()s in flight on one file only
// take turns (tools/io_probe.cpp: 32 MB into a fresh tmpfs file takes 6.2 ms with 1 thread and 6.2 ms with 16), while
// writes to DIFFERENT files do run side by side (22 GB/s with four files).  What does help a fresh file is allocating
// its pages in one go before the first byte is copied: fallocate() of 32 MB takes 1.6 ms and the copy over the
// allocated pages 2.9 ms, against 6.2 ms for write() allocating page by page.  So the strand's first job is ONE fallocate
// of `expectBytes` -- from the moment the file is created, while the input is still being read and coded -- and
// finish(finalBytes) cuts the file to what was really written.
class WriteBehind {
    const PosixFile& file;
    size_t fileOffset, subBytes, expect;
    std::mutex mu;
    std::condition_variable cv;
    const uint8_t* base = nullptr;
    size_t target = 0, written = 0; // bytes of base[] that are final / that are in the file
    bool active = false, allocated = false, failed = false;
    std::string error;
    void strand();

public:
    WriteBehind(const PosixFile& f, size_t fileOffset, size_t subBytes, size_t expectBytes = 0);
    WriteBehind(const WriteBehind&) = delete;
    ~WriteBehind();
    // base[0, upTo) is final (the same base every time): whatever of it is not in the file yet gets written
    void drain(const void* base, size_t upTo);
    // waits until nothing of base[] is being read (no write in flight): the caller may then move or free the buffer and
    // drain() again from its new place -- what is in the file stays there
    void quiesce();
    // waits for everything drained so far; with truncateTo, the file then ends at fileOffset + *truncateTo
    void finish(const size_t* truncateTo = nullptr);
};

} // namespace sela_host
