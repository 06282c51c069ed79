// Constants of the RIRB1 block format; shared by the kernels and the host ABI.
#pragma once
#define RIRB1_TILE_PX 512	   // pixels per tile = 64 lanes x 8 pixels (one 16-byte load per lane)
#define RIRB1_REC_MAX_WORDS 128 // 8 slots x 16 bit-planes (the header lives in a side table)
#define RIRB1_MODE_RAW 0
#define RIRB1_MODE_TEMPORAL 1
#define RIRB1_MODE_LEFT 2
#define RIRB1_DEFAULT_GOP 50	   // reference key-frame cadence, h264.cpp:1662-1665
