/* Umbrella header (reference: libgpujpeg/gpujpeg.h). */
#ifndef GPUJPEG_H
#define GPUJPEG_H
#include "gpujpeg_common.h"
#include "gpujpeg_decoder.h"
#include "gpujpeg_encoder.h"
#include "gpujpeg_type.h"
#include "gpujpeg_version.h"
#endif
