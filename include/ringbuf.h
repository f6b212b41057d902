// ringbuf.h -- ring_buffer<T>: the host staging ring between a sample producer and the
// block processor, with the reference's interface and semantics (libdsp/ringbuf.h:37-142):
//   write(data, len)   all-or-nothing; returns len, or 0 when there is not enough space
//   read(dst, dst_len, conv, calc_src_len)
//                      pulls calc_src_len(dst_len) items through the converting callback
//                      `conv(dst, src, src_len) -> bytes written`, split in two calls when
//                      the span wraps; returns items consumed, or 0 when too few are queued
//   get_space() / get_count() / alloc_buffer(capacity)
// No internal locking: the caller serialises, as examples/bpsk/bpsk.cxx:132-170 does.
// Own implementation (index arithmetic on a monotonically advancing head), not a copy.
#ifndef SFE_DROPIN_RINGBUF_H_
#define SFE_DROPIN_RINGBUF_H_

#include <stddef.h>
#include <string.h>

template <class T>
class ring_buffer
{
public:
    typedef int (*conv_fn)(void *dst, void *src, int src_len);
    typedef int (*len_fn)(int dst_len);

    ring_buffer() : m_store(0), m_cap(0), m_head(0), m_fill(0) {}
    explicit ring_buffer(int capacity) : m_store(0), m_cap(0), m_head(0), m_fill(0) { alloc_buffer(capacity); }
    ~ring_buffer() { delete[] m_store; }

    void alloc_buffer(int capacity)
    {
        delete[] m_store;
        m_store = new T[capacity > 0 ? capacity : 1];
        m_cap = capacity;
        m_head = 0;
        m_fill = 0;
    }

    int get_space() { return m_cap - m_fill; }
    int get_count() { return m_fill; }

    int write(const T *data, int len)
    {
        if (len > get_space()) return 0;
        const int tail = wrap(m_head + m_fill);
        const int first = len < m_cap - tail ? len : m_cap - tail;
        memcpy(m_store + tail, data, (size_t)first * sizeof(T));
        if (first < len) memcpy(m_store, data + first, (size_t)(len - first) * sizeof(T));
        m_fill += len;
        return len;
    }

    int read(void *dst, unsigned dst_len, conv_fn conv, len_fn calc_src_len)
    {
        if (!conv || !calc_src_len) return 0;
        const int want = calc_src_len((int)dst_len);
        if (want > m_fill) return 0;
        const int first = want < m_cap - m_head ? want : m_cap - m_head;
        int written = conv(dst, m_store + m_head, first);
        if (first < want) conv(static_cast<char *>(dst) + written, m_store, want - first);
        m_head = wrap(m_head + want);
        m_fill -= want;
        return want;
    }

private:
    ring_buffer(const ring_buffer &);
    ring_buffer &operator=(const ring_buffer &);
    int wrap(int i) const { return i >= m_cap ? i - m_cap : i; }

    T  *m_store;
    int m_cap;
    int m_head;   // index of the oldest queued item
    int m_fill;   // queued items
};

#endif
