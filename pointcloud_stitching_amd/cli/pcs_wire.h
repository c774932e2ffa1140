// pcs_wire.h — the reference's TCP framing, shared by the host programs.
//
//   frame  = [int32 LE payload_bytes][payload_bytes of 10-byte points]   src/pcs-camera-optimized.cpp:715-720
//   pull   = one byte 'Z' (XYZRGB) from the consumer                      :180-184, src/pcs-multicamera-client.cpp:47, 366-370
//   ports  = 8000+i per camera, 9000 for the stitched cloud               :30, src/pcs-multicamera-client.cpp:42-43
//
// Blocking IPv4 stream sockets like the reference; errors are returned, not exit()ed.
#pragma once
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>

namespace pcs_wire {

constexpr char kPullXYZRGB = 'Z';
constexpr char kPullXYZ = 'Y';          // defined by the reference, only used in its dead code

// loop-until-n read (src/pcs-multicamera-client.cpp:255-268)
inline bool read_n(int fd, void* dst, size_t n)
{
    uint8_t* p = static_cast<uint8_t*>(dst);
    size_t got = 0;
    while (got < n) {
        ssize_t r = ::read(fd, p + got, n - got);
        if (r <= 0) return false;
        got += (size_t)r;
    }
    return true;
}

inline bool write_n(int fd, const void* src, size_t n)
{
    const uint8_t* p = static_cast<const uint8_t*>(src);
    size_t put = 0;
    while (put < n) {
        ssize_t r = ::send(fd, p + put, n - put, MSG_NOSIGNAL);
        if (r <= 0) return false;
        put += (size_t)r;
    }
    return true;
}

// `buffer` holds the 4-byte header followed by the payload (the layout pcs_process_frames produces).
inline bool send_frame(int fd, const int16_t* buffer, int32_t payload_bytes)
{
    return write_n(fd, buffer, (size_t)payload_bytes + sizeof(int32_t));
}

// Reads one frame's payload into dst (capacity in bytes). Returns payload bytes, or -1.
inline int32_t recv_frame(int fd, void* dst, size_t capacity)
{
    int32_t size = 0;
    if (!read_n(fd, &size, sizeof size)) return -1;
    if (size < 0 || (size_t)size > capacity) return -1;
    if (!read_n(fd, dst, (size_t)size)) return -1;
    return size;
}

inline bool send_pull(int fd, char kind = kPullXYZRGB) { return write_n(fd, &kind, 1); }

inline int recv_pull(int fd)      // returns the request byte, or -1 when the peer is gone
{
    char c = 0;
    ssize_t r = ::recv(fd, &c, 1, 0);
    return r == 1 ? (int)(unsigned char)c : -1;
}

inline int listen_on(int port)
{
    int fd = ::socket(AF_INET, SOCK_STREAM, IPPROTO_TCP);
    if (fd < 0) return -1;
    int one = 1;
    ::setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in a;
    std::memset(&a, 0, sizeof a);
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = INADDR_ANY;
    a.sin_port = htons((uint16_t)port);
    if (::bind(fd, (sockaddr*)&a, sizeof a) < 0 || ::listen(fd, 3) < 0) { ::close(fd); return -1; }
    return fd;
}

inline int connect_to(const char* host, int port)
{
    hostent* he = ::gethostbyname(host);
    if (!he) return -1;
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    sockaddr_in a;
    std::memset(&a, 0, sizeof a);
    a.sin_family = AF_INET;
    std::memcpy(&a.sin_addr.s_addr, he->h_addr, (size_t)he->h_length);
    a.sin_port = htons((uint16_t)port);
    if (::connect(fd, (sockaddr*)&a, sizeof a) < 0) { ::close(fd); return -1; }
    int one = 1;
    ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    return fd;
}

}  // namespace pcs_wire
