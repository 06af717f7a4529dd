// tests/qtshim/uvgrtp/media_stream.hh -- TEST INFRASTRUCTURE: the one name of uvgRTP that /root/reference/src/global.h mentions
// (a pointer member of UvgRTPStream); uvgRTP itself is an external dependency of the reference that is absent here.
#pragma once
namespace uvgrtp { class media_stream; }
