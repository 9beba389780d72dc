// hsk_launch.h -- launcher prototypes shared between the kernel translation units and the C-ABI layer.
#pragma once
#include "hsk_dev.h"
#include "hsk_icp_dev.h"

// volume
void launch_integrate(hipStream_t s, void* vol, const float* scaled, const TrackState* st, const VolParams& vp, int W,
                      int H, Intr in, bool count_only, unsigned long long* counter, unsigned* flags,
                      const float* tmax, int2* zint, unsigned* queue, const IcpFinal* icp_final = nullptr,
                      unsigned char* uni = nullptr, const RingOut* early = nullptr);
size_t uniform_bytes(const VolParams& vp);  // lane-block summaries (integrate.hip: hsk_uniform_code)
size_t uniform_lane_bytes(const VolParams& vp);  // ... of which the lane-block bytes; the wave-chunk bytes of the coarse level follow
void launch_rebuild_uniform(hipStream_t s, const void* vol, const VolParams& vp, unsigned char* uni);
void launch_materialize(hipStream_t s, void* vol, const VolParams& vp, unsigned char* uni);  // before anything reads weights
size_t integrate_queue_words(const VolParams& vp);
size_t integrate_queue_counter_words();  // the head of the queue buffer that holds the counters ...
unsigned long long integrate_queue_entries(const unsigned* counter_words);  // ... and their sum, from a host copy of it
unsigned long long integrate_queue_light_entries(const unsigned* counter_words);  // ... and of the light class (free space over holes)
size_t integrate_cflag_offset_bytes(const VolParams& vp);  // the coarse level's verdict bytes inside the zint buffer ...
size_t integrate_chunk_count(const VolParams& vp);         // ... one per wave-chunk
size_t integrate_zint_entries(const VolParams& vp);  // column z ranges + workgroup z ranges (launch_integrate's zint)
void launch_tile_max(hipStream_t s, const float* scaled, int W, int H, float* tmax);
void launch_tile_fine(hipStream_t s, const float* scaled, int W, int H, float* tiles);
void launch_tile_tables(hipStream_t s, int W, int H, float* tiles);  // the window forms + the sparse table, from the raw tables
size_t tile_table_bytes(int W, int H);  // allocation of `tiles` (launch_tile_max + launch_tile_fine fill it)
void launch_rebuild_flags(hipStream_t s, const void* vol, const VolParams& vp, unsigned* flags);
// stored planes [zz0, zz0 + nz) of the volume (64-B blocks, hsk_dev.h: hsk_vox_index) to / from a row-major device array
void launch_vol_to_linear(hipStream_t s, const void* vol, const VolParams& vp, int zz0, int nz, void* lin);
void launch_vol_from_linear(hipStream_t s, void* vol, const VolParams& vp, int zz0, int nz, const void* lin);
void launch_raycast(hipStream_t s, const void* vol, const TrackState* st, const VolParams& vp, int W, int H, Intr in,
                    float* vmap, float* nmap, int* keys, const unsigned* flags, const MapPyramid* pyramid = nullptr,
                    const RingOut* ring = nullptr);
bool raycast_can_fuse_pyramid(const VolParams& vp, int W, int H);
void launch_resolve(hipStream_t s, const int* keys_local, const int* keys_min, const float* vmap, const float* nmap,
                    int* bits, int P);
// the end of a z-slab frame in ONE launch: k_adopt + k_resize_maps2 (+ the report into the host ring) fused (kernels_image.hip)
void launch_adopt_pyramid(hipStream_t s, const int* keys_min, const int* bits, int W, int H, float* v0, float* n0, float* v1, float* n1,
                          float* v2, float* n2, const TrackState* st, const RingOut* ring);
void launch_resolve_push(hipStream_t s, const int* keys_local, const int* keys_min, const float* vmap, const float* nmap,
                         const PushDests& dst, int P);
int extract_warm();   // loads extract.hip's code object (hsk_prepare_readout); a hipError_t
void launch_extract(hipStream_t s, const void* vol, const VolParams& vp, unsigned* row_count,
                    unsigned long long* row_offset, unsigned long long* total, float* xyz, unsigned long long cap,
                    int pass, const unsigned* flags);
size_t hsk_scan_scratch_entries(int nrows);  // entries of a row_offset buffer for nrows rows (the offsets, then the scan's block sums)

// image
void launch_bilateral_scale(hipStream_t s, const uint16_t* src, int W, int H, Intr in, const float* ws, const float* wc,
                            uint16_t* dst, float* scaled, float* tiles);
void launch_scale_depth(hipStream_t s, const uint16_t* src, int W, int H, Intr in, float* scaled);
void launch_pyrdown(hipStream_t s, const uint16_t* src, int W, int H, uint16_t* dst);
void launch_vmap_nmap_pyramid(hipStream_t s, uint16_t* const* depth, const ImgLevel* lv, float* const* vmap,
                              float* const* nmap);
// pyrDown x 2 and the vertex / normal maps of all three levels in one launch (depth[0] is read, depth[1], depth[2] written)
void launch_pyramid_maps(hipStream_t s, uint16_t* const* depth, const ImgLevel* lv, float* const* vmap, float* const* nmap);
void launch_transform_maps(hipStream_t s, const float* vs, const float* ns, int P, const TrackState* st, float* vd,
                           float* nd);
void launch_resize_maps2(hipStream_t s, const float* v0, const float* n0, int W, int H, float* v1, float* n1, float* v2,
                         float* n2, const TrackState* st,
                         const RingOut* ring = nullptr);
int icp_num_blocks(int W, int rows);
void launch_icp_accumulate(hipStream_t s, const float* vcur, const float* ncur, const float* vprev, const float* nprev,
                           int W, int H, Intr in, const TrackState* st, float dist_thresh, float angle_thresh, int row0,
                           int row1, double* partials);
void launch_icp_reduce(hipStream_t s, const double* partials, int nblocks, double* out27);
void launch_icp_update(hipStream_t s, const double* sums27, TrackState* st);
void launch_begin_frame(hipStream_t s, TrackState* st, void* icp_pose_buf);
size_t icp_pose_bytes();
void launch_icp_fused(hipStream_t s, float* const* vcur, float* const* ncur, float* const* vmod, float* const* nmod,
                      const ImgLevel* lv, const int* iters, TrackState* st, float dist_thresh, float angle_thresh,
                      void* pose_buf, double* part_a, double* part_b, IcpFinal* defer_final = nullptr,
                      hipEvent_t* level_events = nullptr);
bool host_solve6(const double* in27, float* x6);
void host_pose_update(float* R, float* t, const float* x6);
void hsk_build_tet_table(TetTable* tt);
int hsk_build_cube_table(CubeTable* ct);  // marching cubes; returns the most triangles of a case (HSK_MC_MAXT)
int hsk_mesh_z_end(const VolParams& vp);
void launch_extract_mesh(hipStream_t s, const void* vol, const VolParams& vp, const TetTable& tt, unsigned* row_count,
                         unsigned long long* row_offset, unsigned long long* total, float* tri, unsigned long long cap, int pass, const unsigned* flags);
void launch_extract_mesh_mc(hipStream_t s, const void* vol, const VolParams& vp, const CubeTable* ct_dev, unsigned* row_count,
                            unsigned long long* row_offset, unsigned long long* total, float* tri, unsigned long long cap, int pass, const unsigned* flags);
