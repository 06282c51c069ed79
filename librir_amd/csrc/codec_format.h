// Constants of the RIRB1 block format; shared by the kernels and the host ABI.
#pragma once
#define RIRB1_TILE_PX 512	   // pixels per tile = 64 lanes x 8 pixels (one 16-byte load per lane)
#define RIRB1_REC_MAX_WORDS 128 // 8 slots x 16 bit-planes (the header lives in a side table)
#define RIRB1_MODE_RAW 0
#define RIRB1_MODE_TEMPORAL 1
#define RIRB1_MODE_LEFT 2
#define RIRB1_DEFAULT_GOP 50	   // reference key-frame cadence, h264.cpp:1662-1665

// Sparse staging slot of one (chunk, tile) segment inside the encoder workspace: worst-case payload plus a pad that
// makes the slot stride an odd multiple of 128 bytes.  Waves write near the same offset of their own slots at the
// same time; with the unpadded stride (gop * 1 KiB, a multiple of 2 KiB for every even gop) all those writes share
// their address bits 7..10 and can land on one L2 / HBM channel group, depending on where the workspace sits.
#ifndef RIRB1_SLOT_PAD_WORDS
#define RIRB1_SLOT_PAD_WORDS 16
#endif
#define RIRB1_SLOT_WORDS(gop) ((int64_t)(gop) * RIRB1_REC_MAX_WORDS + RIRB1_SLOT_PAD_WORDS)

// Control block of the packed form's encoder (rirb1_encode_packed), at the start of its workspace: three 128-byte lines -
// stream cursor, spill cursor, error word - and a spare one.
#define RIRB1_PACKED_CTRL_BYTES 4096
