// Every A/B switch of the kernel dispatchers in ONE table: parsed from the environment once per process (first use), read from then on as a relaxed
// atomic load - no getenv on the launch path.  Tests and benchmarks that must reach a non-default configuration inside one process use the explicit
// override entry point (mis_dispatch_override, include/misamd.h) instead of mutating the environment.
#pragma once

enum MisSwitch {
    SW_CONV_V1, SW_CONV_V3, SW_CONV_NOPP, SW_CONV_PP64, SW_CONV_PPC64, SW_CONV_K3_NO256, SW_CONV_K3_256_MINCIN, SW_CONV_NODMA, SW_CONV_NOWS64,
    SW_CONV3D_BN64V1, SW_CONV_K1V1, SW_CONV_K1NOPERSIST, SW_CONV_K1_NO256,
    SW_CONV_PPC_COLMAJOR, SW_CONV_RS64, SW_CONV_NOPPC, SW_CONV_PPC, SW_CONV_PP_NO256,
    SW_CONV3D_NOPP, SW_CONV3D_PF, SW_CONV3D_ZG, SW_CONV3D_COLMAJOR,
    SW_WGRAD_K1_NARROW, SW_WGRAD_NO_TR, SW_WGRAD_BLOCKS, SW_WGRAD_NOPP, SW_WGRAD_PP_NOWIDE, SW_WGRAD_PP_KSS1, SW_WGRAD3D_NOPP, SW_WGRAD_PP_ROW, SW_WGRAD_PP_NOROW,
    SW_FIRST2D_UNTILED, SW_FIRST3D_UNTILED, SW_UPCONV_BWD_GENERIC, SW_GEMM1_NOPP, SW_CONV_NOPPD, SW_WGRAD_PP_NOSTREAM, SW_WGRAD_K1_NOPP, SW_FIRST3D_NOMFMA, SW_PERSIST_CUS, SW_HEAD_UNFUSED,
    SW_CONV3D_F32_NOPP, SW_WGRAD_F32_NOPP, SW_WGRAD_F32_ROUNDS, SW_CONV_PPS, SW_CONV_PPC2, SW_TILEQ_OFF, SW_CONV3D_NOPF10N4, SW_CONV3D_F32_WIDE,
    SW_COUNT
};

// current value of a switch: the override if one is set, else the environment's value at first use, else the switch's default (0 for the on/off ones)
int mis_sw(MisSwitch k);

// Grid size of the persistent kernels (one block per CU; conv_pp*.hip, gemm1_pp.hip, wgrad_pp.hip): MIS_PERSIST_CUS (default 256 = every CU of an MI355X), clamped to
// [8, 256].  Under data parallelism an RCCL all-reduce kernel runs BESIDE the backward kernels and needs CUs of its own: with 256 one-block-per-CU blocks resident, its
// workgroups wait for a CU to drain, and the last blocks of whichever MFMA kernel shares the device with it finish late.  MIS_PERSIST_CUS=248 (say) leaves 8 CUs free -
// an A/B switch for the first multi-GPU run; single-GPU runs keep 256.
int mis_persist_cus();

// Dynamic tile queue of the persistent kernels (round 5; conv_pp_common.hpp `TileQ`): the first tile of a block is the static one, every further tile is drawn from a
// per-XCD ticket counter, so that a block that finds its CU taken (by an RCCL kernel of the side stream) no longer owns 1 / 256 of the launch - the blocks that do run share
// the tiles, and the late block finds the queue empty.  One 512-byte counter block (8 counters, 64 bytes apart) per STREAM: kernels of one stream never overlap, and the
// block that draws a counter's last ticket stores 0 to it, so the counters are zero again when the kernel ends (graph replays included).  nullptr (pool exhausted, the
// pool could not be allocated because a capture is running, MIS_TILEQ_OFF=1): the kernel falls back to the static stride.
// (A captured launch carries the counter block of the stream it was CAPTURED on: graphs captured on one stream must not be replayed concurrently on two streams - replays on one
// stream, the normal case, are ordered like eager launches.)
unsigned* mis_tile_queue(void* stream);
