/* libgpujpeg ABI version implemented by the MI355X build (tracks reference v0.27.13,
 * libgpujpeg/gpujpeg_version.h.in). */
#ifndef GPUJPEG_VERSION_H
#define GPUJPEG_VERSION_H

#define GPUJPEG_VERSION_MAJOR 0
#define GPUJPEG_VERSION_MINOR 27
#define GPUJPEG_VERSION_PATCH 13

#define GPUJPEG_MK_VERSION_INT(major, minor, patch) ((major) << 16U | (minor) << 8U | (patch))
#define GPUJPEG_VERSION_INT GPUJPEG_MK_VERSION_INT(GPUJPEG_VERSION_MAJOR, GPUJPEG_VERSION_MINOR, GPUJPEG_VERSION_PATCH)
#define LIBGPUJPEG_API_VERSION ((GPUJPEG_VERSION_MAJOR << 8U) | GPUJPEG_VERSION_MINOR)

#endif
