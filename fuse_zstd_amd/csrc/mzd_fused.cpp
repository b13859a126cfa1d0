// mzd_fused -- the read side of fuse-zstd as a multi-threaded daemon on the raw FUSE protocol, with the GPU
// decoder (libmzd.so, include/mzd.h) behind `open`.  No libfuse: the daemon reads
// `fuse_in_header` requests from a descriptor and writes `fuse_out_header` replies to it.  The descriptor is
// /dev/fuse after mount(2) (--mount) or any message-preserving descriptor handed in by the caller (--fd: a
// SOCK_SEQPACKET socketpair in tests/test_fuse_daemon.py, which plays the kernel's part).
//
// What it mirrors of the reference (src/main.rs), read side only:
//   * a regular file `name.zst` of the data directory appears as `name`, directories as they are, other regular
//     files are hidden                                                     (lookup_wrapper :215-260, readdir :307-387)
//   * st_size = xattr user.real_size (8 bytes big-endian), 0 when absent; permissions opened to all
//                                                                          (update_realsize :40-47, getattr :389-405)
//   * inode numbers come from xattr user.ino; a file without one gets the next number of a counter that runs
//     downwards and is kept in xattr user.ino_idx of the data directory    (update_inode :744-754, :719-742)
//   * open decodes the WHOLE file, publishes user.real_size, keeps the bytes until the last release; a second open of
//     the same inode shares them (OpenedFiles::duplicate, src/file.rs:67-102); any decode failure is EFAULT (:467)
//   * read slices the decoded bytes (read_wrapper :495-513)
// (File systems without user xattrs -- tmpfs on older kernels -- are served from in-memory tables instead.)
// Everything that writes (create, write, mkdir, unlink, rename, setattr, ...) answers EROFS: the write path of the
// reference compresses with CPU libzstd and is outside this repository's scope.
//
// Where the GPU pays: `open` requests of all session threads are queued to ONE batcher thread, which waits a short
// time (--batch-us) for more to arrive and decodes them with a single mzd_decode_batch call -- many files per launch.
// There is no CPU decoder in this program: without a usable GPU every open fails (EFAULT), loudly at start-up too.
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <linux/fuse.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mount.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <sys/uio.h>
#include <sys/xattr.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mzd.h"

namespace {

struct Decoded { // the decoded bytes of one inode, shared by all its open handles
    std::vector<uint8_t> bytes;
    int status = MZD_E_DEVICE;
};

struct Options {
    std::string data_dir, mount_point;
    int fd = -1, threads = 4, batch_us = 200, batch_max = 256;
    int ahead = 0;          // decode-ahead: on a miss, also decode the next `ahead` files of the directory (0 = off)
    size_t cache_mb = 256;  // bytes the decode-ahead cache may hold
};

uint64_t be64(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; }
void put_be64(uint8_t* p, uint64_t v) { for (int i = 7; i >= 0; i--) { p[i] = (uint8_t)v; v >>= 8; } }

// ---- the batcher: open() requests of all session threads -> one mzd_decode_batch call
class Batcher {
public:
    Batcher(int wait_us, int max_batch) : wait_us_(wait_us), max_(max_batch), th_([this] { run(); }) {}
    ~Batcher() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); th_.join(); }
    // blocks until `src` is decoded; returns the decoded file (status != 0: failed)
    std::shared_ptr<Decoded> decode(std::vector<uint8_t> src) {
        auto it = std::make_shared<Item>();
        it->src = std::move(src);
        { std::lock_guard<std::mutex> lk(mu_); q_.push_back(it); }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(it->mu);
        it->cv.wait(lk, [&] { return it->done; });
        return it->out;
    }
    // several files at once (decode-ahead): all are queued before anything is waited for, so they share a launch
    std::vector<std::shared_ptr<Decoded>> decode_many(std::vector<std::vector<uint8_t>> srcs) {
        std::vector<std::shared_ptr<Item>> its;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (auto& sv : srcs) { auto it = std::make_shared<Item>(); it->src = std::move(sv); q_.push_back(it); its.push_back(it); }
        }
        cv_.notify_all();
        std::vector<std::shared_ptr<Decoded>> out;
        for (auto& it : its) { std::unique_lock<std::mutex> lk(it->mu); it->cv.wait(lk, [&] { return it->done; }); out.push_back(it->out); }
        return out;
    }
    uint64_t files() const { return files_; }
    uint64_t batches() const { return batches_; }

private:
    struct Item { std::vector<uint8_t> src; std::shared_ptr<Decoded> out; std::mutex mu; std::condition_variable cv; bool done = false; };
    void run() {
        for (;;) {
            std::vector<std::shared_ptr<Item>> batch;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) return;
                // one request is here: give the other session threads a moment to add theirs
#ifdef __SANITIZE_THREAD__ // (GCC 11's ThreadSanitizer does not know pthread_cond_clockwait, which a steady_clock deadline uses: it
                           //  would take the mutex for held across the wait and report double locks that are not there)
                auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(wait_us_);
#else
                auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us_);
#endif
                while ((int)q_.size() < max_ && !stop_ && cv_.wait_until(lk, deadline) != std::cv_status::timeout) {}
                size_t n = std::min<size_t>(q_.size(), (size_t)max_);
                batch.assign(q_.begin(), q_.begin() + (long)n);
                q_.erase(q_.begin(), q_.begin() + (long)n);
            }
            decode_batch(batch);
            for (auto& it : batch) { { std::lock_guard<std::mutex> lk(it->mu); it->done = true; } it->cv.notify_all(); }
        }
    }
    void decode_batch(std::vector<std::shared_ptr<Item>>& batch) {
        std::vector<mzd_job> jobs(batch.size());
        std::vector<size_t> later; // frames without a content size: decoded one by one with a growing buffer
        std::vector<bool> skip(batch.size(), false);
        for (size_t i = 0; i < batch.size(); i++) {
            auto& it = *batch[i];
            it.out = std::make_shared<Decoded>();
            memset(&jobs[i], 0, sizeof(mzd_job));
            uint64_t want = mzd_content_size(it.src.data(), it.src.size());
            if (want == MZD_CONTENTSIZE_ERROR) { it.out->status = MZD_E_CORRUPT; skip[i] = true; continue; } // not a zstd file
            if (want == MZD_CONTENTSIZE_UNKNOWN) { later.push_back(i); skip[i] = true; continue; }
            // (a header may promise anything: what the file's bytes can regenerate is bounded -- a block gives <= 128 KiB and costs >= 4 bytes)
            if (want > ((uint64_t)it.src.size() / 4 + 1) * (128u << 10)) { it.out->status = MZD_E_CORRUPT; skip[i] = true; continue; }
            try { it.out->bytes.resize((size_t)want); } catch (const std::exception&) { it.out->status = MZD_E_DEVICE; skip[i] = true; continue; }
            jobs[i].src = it.src.data(); jobs[i].src_len = it.src.size();
            jobs[i].dst = it.out->bytes.data(); jobs[i].dst_cap = it.out->bytes.size();
        }
        std::vector<mzd_job> run;
        std::vector<size_t> idx;
        for (size_t i = 0; i < batch.size(); i++)
            if (!skip[i]) { run.push_back(jobs[i]); idx.push_back(i); }
        if (!run.empty()) {
            int rc = mzd_decode_batch(run.data(), run.size());
            for (size_t k = 0; k < run.size(); k++) {
                auto& o = *batch[idx[k]]->out;
                o.status = rc != MZD_OK ? rc : run[k].status;
                if (o.status == MZD_OK) { o.bytes.resize(run[k].out_len); files_++; } else o.bytes.clear();
            }
            batches_++;
        }
        for (size_t i : later) {
            auto& it = *batch[i];
            size_t cap = std::max<size_t>(it.src.size() * 8, 1u << 20);
            const uint64_t cap_max = ((uint64_t)it.src.size() / 4 + 1) * (128u << 10);
            for (;;) {
                try { it.out->bytes.resize(cap); } catch (const std::exception&) { it.out->status = MZD_E_DEVICE; it.out->bytes.clear(); break; }
                size_t n = 0;
                int rc = mzd_decode(it.src.data(), it.src.size(), it.out->bytes.data(), cap, &n);
                if (rc == MZD_E_DSTSIZE && cap < cap_max) { cap = (size_t)std::min<uint64_t>((uint64_t)cap * 4, cap_max); continue; }
                it.out->status = rc;
                it.out->bytes.resize(rc == MZD_OK ? n : 0);
                if (rc == MZD_OK) files_++;
                break;
            }
            batches_++;
        }
    }
    int wait_us_, max_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::shared_ptr<Item>> q_;
    bool stop_ = false;
    std::atomic<uint64_t> files_{0}, batches_{0};
    std::thread th_;
};

// ---- the file system state
class Fs {
public:
    explicit Fs(const Options& o) : opt_(o), batcher_(o.batch_us, o.batch_max) {
        uint8_t b[8];
        if (getxattr(opt_.data_dir.c_str(), "user.ino_idx", b, 8) == 8) ino_idx_ = be64(b);
        paths_[FUSE_ROOT_ID] = "";
    }
    const Options& opt() const { return opt_; }
    Batcher& batcher() { return batcher_; }

    // path of an inode relative to the data directory ("" = the directory itself); false: unknown inode
    bool path_of(uint64_t ino, std::string* rel) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = paths_.find(ino);
        if (it == paths_.end()) return false;
        *rel = it->second;
        return true;
    }
    std::string abs(const std::string& rel) const { return rel.empty() ? opt_.data_dir : opt_.data_dir + "/" + rel; }

    // the inode number of a backing file: xattr user.ino, else a new one from the counter (reference update_inode)
    uint64_t inode_for(const std::string& rel) {
        const std::string p = abs(rel);
        uint8_t b[8];
        std::lock_guard<std::mutex> lk(mu_);
        auto known = inos_.find(rel);
        if (known != inos_.end()) return known->second;
        uint64_t ino;
        if (getxattr(p.c_str(), "user.ino", b, 8) == 8) ino = be64(b);
        else {
            ino = ino_idx_;
            ino_idx_ = ino_idx_ - 1 <= FUSE_ROOT_ID ? UINT64_MAX : ino_idx_ - 1; // (never wraps in practice, reference :722-727)
            put_be64(b, ino_idx_);
            (void)setxattr(opt_.data_dir.c_str(), "user.ino_idx", b, 8, 0);
            put_be64(b, ino);
            (void)setxattr(p.c_str(), "user.ino", b, 8, 0); // without xattr support the number lives in `inos_` only
        }
        inos_[rel] = ino;
        paths_[ino] = rel;
        return ino;
    }

    // attributes as the mount shows them; 0 or an errno
    int attr_of(uint64_t ino, const std::string& rel, fuse_attr* a) {
        struct stat st;
        if (stat(abs(rel).c_str(), &st) != 0) return errno;
        memset(a, 0, sizeof(*a));
        a->ino = ino;
        a->blocks = (uint64_t)st.st_blocks; a->atime = (uint64_t)st.st_atime; a->mtime = (uint64_t)st.st_mtime; a->ctime = (uint64_t)st.st_ctime;
        a->nlink = (uint32_t)st.st_nlink; a->uid = st.st_uid; a->gid = st.st_gid; a->rdev = (uint32_t)st.st_rdev; a->blksize = (uint32_t)st.st_blksize;
        if (S_ISDIR(st.st_mode)) { a->mode = S_IFDIR | 0777; a->size = (uint64_t)st.st_size; }
        else if (S_ISREG(st.st_mode)) {
            a->mode = S_IFREG | 0666;
            uint8_t b[8];
            if (getxattr(abs(rel).c_str(), "user.real_size", b, 8) == 8) a->size = be64(b); // (reference: 0 when the file was never opened)
            else { std::lock_guard<std::mutex> lk(mu_); auto it = sizes_.find(ino); a->size = it == sizes_.end() ? 0 : it->second; }
        } else return ENOENT;
        return 0;
    }

    // lookup of `name` in directory `parent`: regular files are stored as name.zst
    int lookup(uint64_t parent, const std::string& name, uint64_t* ino, fuse_attr* a) {
        std::string dir;
        if (!path_of(parent, &dir)) return ENOENT;
        for (int pass = 0; pass < 2; pass++) {
            const std::string rel = (dir.empty() ? "" : dir + "/") + name + (pass == 0 ? ".zst" : "");
            struct stat st;
            if (stat(abs(rel).c_str(), &st) != 0) continue;
            if (pass == 0 ? !S_ISREG(st.st_mode) : !S_ISDIR(st.st_mode)) continue;
            *ino = inode_for(rel);
            return attr_of(*ino, rel, a);
        }
        return ENOENT;
    }

    // open: decode on first open, share afterwards.  Returns 0 + fh or an errno.
    int open(uint64_t ino, uint64_t* fh) {
        std::string rel;
        if (!path_of(ino, &rel)) return ENOENT;
        std::shared_ptr<Decoded> d;
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = by_ino_.find(ino);
            if (it != by_ino_.end()) d = it->second.lock();
        }
        if (!d) d = cache_take(ino); // decoded ahead of this open
        if (!d) {
            // the file itself, and -- with decode-ahead -- the next files of its directory: sequential readers (fio's
            // parallel-files pattern, tar, grep -r) will ask for them next, and a launch is cheaper per file the more it holds
            std::vector<std::string> rels{rel};
            if (opt_.ahead > 0) for (auto& r : followers(rel, (size_t)opt_.ahead)) rels.push_back(r);
            std::vector<std::vector<uint8_t>> srcs;
            std::vector<std::string> got;
            for (size_t i = 0; i < rels.size(); i++) {
                std::vector<uint8_t> src;
                int e = slurp(abs(rels[i]), &src);
                if (e) { if (i == 0) return e; continue; }
                srcs.push_back(std::move(src)); got.push_back(rels[i]);
            }
            auto outs = batcher_.decode_many(std::move(srcs));
            d = outs[0];
            for (size_t i = 1; i < outs.size(); i++)
                if (outs[i]->status == MZD_OK) cache_put(inode_for(got[i]), outs[i]);
            if (d->status != MZD_OK) return EFAULT; // reference src/main.rs:467: every decode failure
        }
        {
            uint8_t b[8];
            put_be64(b, d->bytes.size());
            (void)setxattr(abs(rel).c_str(), "user.real_size", b, 8, 0);
            std::lock_guard<std::mutex> lk(mu_);
            sizes_[ino] = d->bytes.size();
            auto it = by_ino_.find(ino);
            std::shared_ptr<Decoded> other = it != by_ino_.end() ? it->second.lock() : nullptr;
            if (other) d = other; // another thread decoded the same inode meanwhile: share its bytes
            else by_ino_[ino] = d;
        }
        std::lock_guard<std::mutex> lk(mu_);
        *fh = next_fh_++;
        handles_[*fh] = d;
        return 0;
    }
    static int slurp(const std::string& path, std::vector<uint8_t>* out) {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return errno;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { int e = errno ? errno : EISDIR; close(fd); return e; }
        out->resize((size_t)st.st_size);
        size_t got = 0;
        while (got < out->size()) { ssize_t r = ::read(fd, out->data() + got, out->size() - got); if (r <= 0) break; got += (size_t)r; }
        close(fd);
        return got == out->size() ? 0 : EIO;
    }
    // the next `n` .zst files after `rel` in its directory (name order) that are neither open nor cached
    std::vector<std::string> followers(const std::string& rel, size_t n) {
        const size_t slash = rel.rfind('/');
        const std::string dir = slash == std::string::npos ? "" : rel.substr(0, slash), base = slash == std::string::npos ? rel : rel.substr(slash + 1);
        std::vector<std::string> names;
        if (DIR* dp = opendir(abs(dir).c_str())) {
            while (dirent* de = ::readdir(dp)) {
                std::string nm = de->d_name;
                if (nm.size() > 4 && nm.compare(nm.size() - 4, 4, ".zst") == 0 && nm > base) names.push_back(nm);
            }
            closedir(dp);
        }
        std::sort(names.begin(), names.end());
        std::vector<std::string> out;
        for (auto& nm : names) {
            if (out.size() >= n) break;
            const std::string r = (dir.empty() ? "" : dir + "/") + nm;
            const uint64_t ino = inode_for(r);
            std::lock_guard<std::mutex> lk(mu_);
            auto it = by_ino_.find(ino);
            if ((it != by_ino_.end() && !it->second.expired()) || cache_idx_.count(ino)) continue;
            out.push_back(r);
        }
        return out;
    }
    std::shared_ptr<Decoded> cache_take(uint64_t ino) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = cache_idx_.find(ino);
        if (it == cache_idx_.end()) return nullptr;
        auto d = it->second->second;
        cache_bytes_ -= d->bytes.size();
        cache_.erase(it->second);
        cache_idx_.erase(it);
        hits_++;
        return d;
    }
    void cache_put(uint64_t ino, std::shared_ptr<Decoded> d) {
        std::lock_guard<std::mutex> lk(mu_);
        if (cache_idx_.count(ino)) return;
        cache_.emplace_front(ino, d);
        cache_idx_[ino] = cache_.begin();
        cache_bytes_ += d->bytes.size();
        while (cache_bytes_ > opt_.cache_mb * (1u << 20) && cache_.size() > 1) { // oldest out
            auto& last = cache_.back();
            cache_bytes_ -= last.second->bytes.size();
            cache_idx_.erase(last.first);
            cache_.pop_back();
        }
    }
public:
    uint64_t hits() const { return hits_; }
    std::shared_ptr<Decoded> handle(uint64_t fh) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = handles_.find(fh);
        return it == handles_.end() ? nullptr : it->second;
    }
    void release(uint64_t fh) { std::lock_guard<std::mutex> lk(mu_); handles_.erase(fh); }
    void forget(uint64_t) {} // paths stay known: the table is small and a later lookup finds the same number anyway

private:
    Options opt_;
    Batcher batcher_;
    std::mutex mu_;
    std::map<uint64_t, std::string> paths_;
    std::map<std::string, uint64_t> inos_;
    std::map<uint64_t, uint64_t> sizes_;
    std::map<uint64_t, std::weak_ptr<Decoded>> by_ino_;
    std::map<uint64_t, std::shared_ptr<Decoded>> handles_;
    uint64_t next_fh_ = 1, ino_idx_ = UINT64_MAX - 1;
    std::list<std::pair<uint64_t, std::shared_ptr<Decoded>>> cache_; // decoded ahead, not opened yet (newest first)
    std::map<uint64_t, std::list<std::pair<uint64_t, std::shared_ptr<Decoded>>>::iterator> cache_idx_;
    size_t cache_bytes_ = 0;
    std::atomic<uint64_t> hits_{0};
};

// ---- the session: requests in, replies out
class Session {
public:
    Session(Fs& fs, int fd) : fs_(fs), fd_(fd) {}
    void serve() {
        std::vector<uint8_t> buf(FUSE_MIN_READ_BUFFER + (1u << 20));
        while (!done_) {
            ssize_t n = ::read(fd_, buf.data(), buf.size());
            if (n < 0 && (errno == EINTR || errno == EAGAIN || errno == ENOENT)) continue;
            if (n <= 0) break; // unmounted / peer closed
            if (n < (ssize_t)sizeof(fuse_in_header)) continue; // (not a request: nothing to answer)
            dispatch(buf.data(), (size_t)n);
        }
        done_ = true;
    }

private:
    void reply(uint64_t unique, int err, const void* body = nullptr, size_t len = 0) {
        fuse_out_header h;
        h.len = (uint32_t)(sizeof(h) + (err ? 0 : len)); h.error = -err; h.unique = unique;
        iovec iov[2] = {{&h, sizeof(h)}, {const_cast<void*>(body), err ? 0 : len}};
        std::lock_guard<std::mutex> lk(wmu_); // one message at a time on a shared socket
        ssize_t w = writev(fd_, iov, (body && !err && len) ? 2 : 1);
        (void)w;
    }
    void entry_out(uint64_t unique, uint64_t ino, const fuse_attr& a) {
        fuse_entry_out e;
        memset(&e, 0, sizeof(e));
        e.nodeid = ino; e.generation = 0; e.entry_valid = 1; e.attr_valid = 1; e.attr = a; // TTL 1 s (reference :25)
        reply(unique, 0, &e, sizeof(e));
    }
    void dispatch(const uint8_t* p, size_t n) {
        const fuse_in_header* in = reinterpret_cast<const fuse_in_header*>(p);
        const uint8_t* arg = p + sizeof(*in);
        const size_t alen = n - sizeof(*in);
        // the fixed part of a request's body must be there (what arrived counts, not what the header claims)
        size_t need = 0;
        switch (in->opcode) {
        case FUSE_INIT: need = 8; break; // major, minor (7.5 and older send no more)
        case FUSE_GETATTR: need = 0; break; // (the body is optional before 7.9 and unused here)
        case FUSE_OPEN: case FUSE_OPENDIR: need = sizeof(fuse_open_in); break;
        case FUSE_READ: case FUSE_READDIR: need = offsetof(fuse_read_in, read_flags); break; // fh, offset, size (7.8 and older end there)
        case FUSE_RELEASE: need = sizeof(uint64_t); break; // fh
        case FUSE_LOOKUP: need = 1; break;
        default: break;
        }
        if (alen < need) { reply(in->unique, EINVAL); return; }
        switch (in->opcode) {
        case FUSE_INIT: {
            fuse_init_in iibuf;
            memset(&iibuf, 0, sizeof(iibuf));
            memcpy(&iibuf, arg, std::min(alen, sizeof(iibuf)));
            const fuse_init_in* ii = &iibuf;
            fuse_init_out o;
            memset(&o, 0, sizeof(o));
            o.major = FUSE_KERNEL_VERSION;
            o.minor = ii->minor < FUSE_KERNEL_MINOR_VERSION ? ii->minor : FUSE_KERNEL_MINOR_VERSION;
            o.max_readahead = ii->max_readahead; o.flags = 0; o.max_background = 64; o.congestion_threshold = 48;
            o.max_write = 1u << 17; o.time_gran = 1;
            if (ii->major != FUSE_KERNEL_VERSION) { reply(in->unique, EPROTO); break; }
            reply(in->unique, 0, &o, ii->minor < 23 ? 24 : sizeof(o));
            break;
        }
        case FUSE_DESTROY: reply(in->unique, 0); done_ = true; shutdown(fd_, SHUT_RD); break; // (wakes the other session threads on a socket; /dev/fuse reads fail after the unmount)
        case FUSE_FORGET: fs_.forget(in->nodeid); break;       // no reply
        case FUSE_BATCH_FORGET: break;                         // no reply
        case FUSE_INTERRUPT: break;                            // no reply
        case FUSE_LOOKUP: {
            std::string name(reinterpret_cast<const char*>(arg), strnlen(reinterpret_cast<const char*>(arg), alen));
            uint64_t ino; fuse_attr a;
            int e = fs_.lookup(in->nodeid, name, &ino, &a);
            if (e) reply(in->unique, e); else entry_out(in->unique, ino, a);
            break;
        }
        case FUSE_GETATTR: {
            std::string rel; fuse_attr_out o;
            memset(&o, 0, sizeof(o));
            if (!fs_.path_of(in->nodeid, &rel)) { reply(in->unique, ENOENT); break; }
            int e = fs_.attr_of(in->nodeid, rel, &o.attr);
            o.attr_valid = 1;
            if (e) reply(in->unique, e); else reply(in->unique, 0, &o, sizeof(o));
            break;
        }
        case FUSE_OPENDIR: {
            std::string rel;
            if (!fs_.path_of(in->nodeid, &rel)) { reply(in->unique, ENOENT); break; }
            struct stat st;
            if (stat(fs_.abs(rel).c_str(), &st) != 0) { reply(in->unique, errno); break; }
            if (!S_ISDIR(st.st_mode)) { reply(in->unique, ENOTDIR); break; }
            fuse_open_out o; memset(&o, 0, sizeof(o));
            reply(in->unique, 0, &o, sizeof(o));
            break;
        }
        case FUSE_READDIR: readdir(in, reinterpret_cast<const fuse_read_in*>(arg)); break;
        case FUSE_RELEASEDIR: reply(in->unique, 0); break;
        case FUSE_OPEN: {
            const fuse_open_in* oi = reinterpret_cast<const fuse_open_in*>(arg);
            if ((oi->flags & O_ACCMODE) != O_RDONLY) { reply(in->unique, EROFS); break; }
            uint64_t fh = 0;
            int e = fs_.open(in->nodeid, &fh);
            if (e) { reply(in->unique, e); break; }
            fuse_open_out o; memset(&o, 0, sizeof(o));
            o.fh = fh; o.open_flags = FOPEN_KEEP_CACHE;
            reply(in->unique, 0, &o, sizeof(o));
            break;
        }
        case FUSE_READ: {
            const fuse_read_in* ri = reinterpret_cast<const fuse_read_in*>(arg);
            auto d = fs_.handle(ri->fh);
            if (!d) { reply(in->unique, ENOENT); break; } // reference :505
            size_t off = (size_t)std::min<uint64_t>(ri->offset, d->bytes.size());
            size_t len = std::min<size_t>(ri->size, d->bytes.size() - off); // short at the end of the file
            reply(in->unique, 0, d->bytes.data() + off, len);
            break;
        }
        case FUSE_FLUSH: reply(in->unique, 0); break;
        case FUSE_FSYNC: reply(in->unique, 0); break;
        case FUSE_RELEASE: {
            const fuse_release_in* ri = reinterpret_cast<const fuse_release_in*>(arg);
            fs_.release(ri->fh);
            reply(in->unique, 0);
            break;
        }
        case FUSE_STATFS: {
            struct statvfs sv; fuse_statfs_out o;
            memset(&o, 0, sizeof(o));
            if (statvfs(fs_.opt().data_dir.c_str(), &sv) == 0) {
                o.st.blocks = sv.f_blocks; o.st.bfree = sv.f_bfree; o.st.bavail = sv.f_bavail; o.st.files = sv.f_files; o.st.ffree = sv.f_ffree;
                o.st.bsize = (uint32_t)sv.f_bsize; o.st.namelen = (uint32_t)sv.f_namemax; o.st.frsize = (uint32_t)sv.f_frsize;
            }
            reply(in->unique, 0, &o, sizeof(o));
            break;
        }
        case FUSE_ACCESS: reply(in->unique, 0); break;
        case FUSE_SETATTR: case FUSE_MKNOD: case FUSE_MKDIR: case FUSE_UNLINK: case FUSE_RMDIR: case FUSE_RENAME: case FUSE_RENAME2:
        case FUSE_LINK: case FUSE_SYMLINK: case FUSE_WRITE: case FUSE_CREATE: case FUSE_SETXATTR: case FUSE_REMOVEXATTR: case FUSE_FALLOCATE:
            reply(in->unique, EROFS); // the write path is not part of this daemon
            break;
        default: reply(in->unique, ENOSYS); break;
        }
    }
    void readdir(const fuse_in_header* in, const fuse_read_in* ri) {
        std::string rel;
        if (!fs_.path_of(in->nodeid, &rel)) { reply(in->unique, ENOENT); return; }
        DIR* dp = opendir(fs_.abs(rel).c_str());
        if (!dp) { reply(in->unique, errno == ENOTDIR ? ENOTDIR : errno); return; }
        std::vector<uint8_t> out;
        uint64_t idx = 0; // position in the (filtered) listing; the offset of an entry is the index of the next one
        while (dirent* de = ::readdir(dp)) {
            std::string nm = de->d_name;
            if (nm == "." || nm == "..") continue;
            std::string child = (rel.empty() ? "" : rel + "/") + nm;
            struct stat st;
            if (stat(fs_.abs(child).c_str(), &st) != 0) continue;
            std::string shown;
            uint32_t type;
            if (S_ISREG(st.st_mode)) {
                if (nm.size() <= 4 || nm.compare(nm.size() - 4, 4, ".zst") != 0) continue; // hidden (reference :339-343, no convert mode)
                shown = nm.substr(0, nm.size() - 4); type = DT_REG;
            } else if (S_ISDIR(st.st_mode)) { shown = nm; type = DT_DIR; }
            else continue;
            idx++;
            if (idx <= ri->offset) continue;
            const size_t entlen = FUSE_NAME_OFFSET + shown.size(), padded = FUSE_DIRENT_ALIGN(entlen);
            if (out.size() + padded > ri->size) break;
            const size_t at = out.size();
            out.resize(at + padded, 0);
            fuse_dirent* d = reinterpret_cast<fuse_dirent*>(out.data() + at);
            d->ino = fs_.inode_for(child); d->off = idx; d->namelen = (uint32_t)shown.size(); d->type = type;
            memcpy(d->name, shown.data(), shown.size());
        }
        closedir(dp);
        reply(in->unique, 0, out.data(), out.size());
    }
    Fs& fs_;
    int fd_;
    std::mutex wmu_;
    std::atomic<bool> done_{false};
};

int usage() {
    fprintf(stderr, "usage: mzd_fused --data-dir DIR (--mount DIR | --fd N) [--threads T] [--batch-us U] [--batch-max B] [--ahead N] [--cache-mb M]\n");
    return 2;
}

} // namespace

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0); // before the first HIP call: the host path overlaps four kernel streams and two copy streams (include/mzd.h); the caller's own setting wins
    Options o;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--data-dir") o.data_dir = val();
        else if (a == "--mount") o.mount_point = val();
        else if (a == "--fd") o.fd = atoi(val());
        else if (a == "--threads") o.threads = atoi(val());
        else if (a == "--batch-us") o.batch_us = atoi(val());
        else if (a == "--batch-max") o.batch_max = atoi(val());
        else if (a == "--ahead") o.ahead = atoi(val());
        else if (a == "--cache-mb") o.cache_mb = (size_t)atol(val());
        else return usage();
    }
    if (o.data_dir.empty() || (o.mount_point.empty() && o.fd < 0)) return usage();
    if (o.threads < 1) o.threads = 1;
    if (o.batch_max < 1) o.batch_max = 1;
    int rc = mzd_init(nullptr, 0);
    if (rc != MZD_OK) fprintf(stderr, "mzd_fused: no usable MI355X (%s): every open will fail with EFAULT -- there is no CPU decoder here\n", mzd_strerror(rc));
    int fd = o.fd;
    if (fd < 0) {
        fd = open("/dev/fuse", O_RDWR | O_CLOEXEC);
        if (fd < 0) { perror("mzd_fused: /dev/fuse"); return 1; }
        struct stat st;
        if (stat(o.mount_point.c_str(), &st) != 0) { perror("mzd_fused: mount point"); return 1; }
        char opts[160];
        snprintf(opts, sizeof opts, "fd=%d,rootmode=%o,user_id=%u,group_id=%u,allow_other,default_permissions", fd, st.st_mode & S_IFMT, getuid(), getgid());
        if (mount("mzd_fused", o.mount_point.c_str(), "fuse.mzd_fused", MS_NOSUID | MS_NODEV | MS_RDONLY, opts) != 0) { perror("mzd_fused: mount"); return 1; }
    }
    {
        Fs fs(o);
        Session s(fs, fd);
        std::vector<std::thread> th;
        for (int t = 0; t < o.threads; t++) th.emplace_back([&] { s.serve(); });
        for (auto& t : th) t.join();
        fprintf(stderr, "mzd_fused: %llu files decoded in %llu batches, %llu opens served from decode-ahead\n", (unsigned long long)fs.batcher().files(),
                (unsigned long long)fs.batcher().batches(), (unsigned long long)fs.hits());
    }
    if (o.fd < 0) umount2(o.mount_point.c_str(), MNT_DETACH);
    mzd_shutdown();
    return 0;
}
