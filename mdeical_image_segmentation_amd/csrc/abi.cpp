// Layout of every struct that crosses the C ABI, as THIS build of the library sees it: the Python side keeps hand-written ctypes / numpy mirrors of these
// structs (mdeical_image_segmentation_amd/_lib.py, ops.py), and a field added on one side only - or two fields swapped - is a silent wild pointer on the device (the round-4
// fault came through exactly this class of bug).  mis_abi_layout lets the CPU test suite compare sizeof, field names, offsets and sizes of every mirror with the C truth.
#include <stddef.h>
#include <string.h>

#include "common.hpp"

namespace {
struct AbiField {
    const char* name;
    size_t offset, size;
};
struct AbiStruct {
    const char* name;
    size_t size;
    const AbiField* fields;
    int n;
};
#define F(S, f) {#f, offsetof(S, f), sizeof(((S*)nullptr)->f)}

const AbiField conv_fields[] = {
    F(MisConvDesc, dtype), F(MisConvDesc, ksize), F(MisConvDesc, N), F(MisConvDesc, D), F(MisConvDesc, H), F(MisConvDesc, W), F(MisConvDesc, is3d), F(MisConvDesc, Cin),
    F(MisConvDesc, Cout), F(MisConvDesc, x0), F(MisConvDesc, x0_ld), F(MisConvDesc, x0_D), F(MisConvDesc, x0_H), F(MisConvDesc, x0_W), F(MisConvDesc, x1), F(MisConvDesc, x1_ld),
    F(MisConvDesc, x1_D), F(MisConvDesc, x1_H), F(MisConvDesc, x1_W), F(MisConvDesc, Cin0), F(MisConvDesc, in_scale), F(MisConvDesc, in_shift), F(MisConvDesc, w),
    F(MisConvDesc, bias), F(MisConvDesc, relu), F(MisConvDesc, mask), F(MisConvDesc, mask_ld), F(MisConvDesc, y0), F(MisConvDesc, y0_ld), F(MisConvDesc, y0_mode),
    F(MisConvDesc, y1), F(MisConvDesc, y1_ld), F(MisConvDesc, y1_mode), F(MisConvDesc, Cout0), F(MisConvDesc, relu_bits), F(MisConvDesc, mask_bits), F(MisConvDesc, gn_p),
    F(MisConvDesc, gn_q), F(MisConvDesc, gn_r), F(MisConvDesc, gn_ld), F(MisConvDesc, gn_relu), F(MisConvDesc, st_mode), F(MisConvDesc, st_x0), F(MisConvDesc, st_x0_ld),
    F(MisConvDesc, st_x1), F(MisConvDesc, st_x1_ld), F(MisConvDesc, st_c0), F(MisConvDesc, st_up), F(MisConvDesc, st_part),
};
const AbiField wgrad_fields[] = {
    F(MisWgradDesc, dtype), F(MisWgradDesc, ksize), F(MisWgradDesc, N), F(MisWgradDesc, D), F(MisWgradDesc, H), F(MisWgradDesc, W), F(MisWgradDesc, is3d), F(MisWgradDesc, Cin),
    F(MisWgradDesc, Cout), F(MisWgradDesc, x0), F(MisWgradDesc, x0_ld), F(MisWgradDesc, x0_D), F(MisWgradDesc, x0_H), F(MisWgradDesc, x0_W), F(MisWgradDesc, x1),
    F(MisWgradDesc, x1_ld), F(MisWgradDesc, x1_D), F(MisWgradDesc, x1_H), F(MisWgradDesc, x1_W), F(MisWgradDesc, Cin0), F(MisWgradDesc, in_scale), F(MisWgradDesc, in_shift),
    F(MisWgradDesc, dy), F(MisWgradDesc, dy_ld), F(MisWgradDesc, workspace), F(MisWgradDesc, workspace_bytes), F(MisWgradDesc, dw), F(MisWgradDesc, dw_layout),
    F(MisWgradDesc, alpha), F(MisWgradDesc, dbias), F(MisWgradDesc, reduce_stream), F(MisWgradDesc, dw_per_sample), F(MisWgradDesc, dbias_per_sample), F(MisWgradDesc, defer),
};
const AbiField red_fields[] = {
    F(MisWgradReduceItem, partial), F(MisWgradReduceItem, dw), F(MisWgradReduceItem, bias_partial), F(MisWgradReduceItem, dbias), F(MisWgradReduceItem, nsplit),
    F(MisWgradReduceItem, TT), F(MisWgradReduceItem, Cin), F(MisWgradReduceItem, Cout), F(MisWgradReduceItem, dw_layout), F(MisWgradReduceItem, alpha),
};
const AbiField head_fields[] = {
    F(MisHeadDesc, dtype), F(MisHeadDesc, loss), F(MisHeadDesc, npix_per_image), F(MisHeadDesc, N), F(MisHeadDesc, Cfeat), F(MisHeadDesc, C), F(MisHeadDesc, y),
    F(MisHeadDesc, y_ld), F(MisHeadDesc, w), F(MisHeadDesc, b), F(MisHeadDesc, labels), F(MisHeadDesc, logits), F(MisHeadDesc, argmax), F(MisHeadDesc, workspace),
    F(MisHeadDesc, workspace_bytes), F(MisHeadDesc, loss_out), F(MisHeadDesc, dy), F(MisHeadDesc, dy_ld), F(MisHeadDesc, dw), F(MisHeadDesc, db), F(MisHeadDesc, grad_scale),
    F(MisHeadDesc, alpha), F(MisHeadDesc, beta), F(MisHeadDesc, phase),
};
const AbiField pack_fields[] = {
    F(MisPackItem, w), F(MisPackItem, w_fwd), F(MisPackItem, w_dgrad), F(MisPackItem, rows), F(MisPackItem, cols), F(MisPackItem, taps), F(MisPackItem, kind),
};
const AbiField pack2_fields[] = {
    F(MisPackItem2, w), F(MisPackItem2, w_fwd), F(MisPackItem2, w_dgrad), F(MisPackItem2, rows), F(MisPackItem2, cols), F(MisPackItem2, taps), F(MisPackItem2, kind),
    F(MisPackItem2, blk0), F(MisPackItem2, nbx),
};
#define S(T, arr) {#T, sizeof(T), arr, (int)(sizeof(arr) / sizeof(arr[0]))}
const AbiStruct g_structs[] = {
    S(MisConvDesc, conv_fields), S(MisWgradDesc, wgrad_fields), S(MisWgradReduceItem, red_fields), S(MisHeadDesc, head_fields), S(MisPackItem, pack_fields),
    S(MisPackItem2, pack2_fields),
};
}   // namespace

extern "C" int mis_abi_struct_count(void) { return (int)(sizeof(g_structs) / sizeof(g_structs[0])); }

extern "C" const char* mis_abi_struct_name(int index) {
    if (index < 0 || index >= mis_abi_struct_count()) return nullptr;
    return g_structs[index].name;
}

extern "C" int mis_abi_layout(const char* struct_name, size_t* size, const char** names, size_t* offsets, size_t* sizes, int max_fields) {
    if (struct_name == nullptr) {
        mis_set_error("mis_abi_layout: null name");
        return MIS_EINVAL;
    }
    for (const AbiStruct& s : g_structs)
        if (strcmp(s.name, struct_name) == 0) {
            if (size != nullptr) *size = s.size;
            for (int i = 0; i < s.n && i < max_fields; ++i) {
                if (names != nullptr) names[i] = s.fields[i].name;
                if (offsets != nullptr) offsets[i] = s.fields[i].offset;
                if (sizes != nullptr) sizes[i] = s.fields[i].size;
            }
            return s.n;
        }
    mis_set_error("mis_abi_layout: unknown struct '%s'", struct_name);
    return MIS_EINVAL;
}
