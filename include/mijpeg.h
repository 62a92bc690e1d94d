/*
 * mijpeg.h -- C ABI of the MI355X-native JPEG block-decode path (libjpeg_amd/libmijpeg.so).
 *
 * This is the drop-in boundary for the hot path of thorfdbg/libjpeg:
 *
 *      JPEG::Read            (interface/jpeg.cpp:205-354)   -> mijpeg_set_input + mijpeg_read_header
 *                                                              + mijpeg_decode_coefficients (+ upload)
 *      JPEG::GetInformation  (interface/jpeg.cpp:822-957)   -> mijpeg_info (filled by read_header)
 *      JPEG::DisplayRectangle(interface/jpeg.cpp:694-722)   -> mijpeg_reconstruct_rect
 *          = Image::ReconstructRegion (codestream/image.cpp:1087-1123)
 *          = BlockBitmapRequester::ReconstructRegion (control/blockbitmaprequester.cpp:1249-1272)
 *      JPEG::LastError       (interface/jpeg.cpp:959-968)   -> mijpeg_last_error
 *
 * The reference has no C ABI (it exports the C++ class JPEG); the source-compatible class JPEG in
 * libjpeg_amd/csrc/interface/ is implemented on top of these entry points, see INTEGRATION.md.
 *
 * Conventions: extern "C", plain pointers and sizes, int return codes (0 = ok, negative = the
 * reference's JPGERR_* value, interface/parameters.hpp:1156-1228), no exceptions cross the
 * boundary, the library never takes ownership of pixel memory.  hipStream_t is passed as void*.
 */
#ifndef MIJPEG_H
#define MIJPEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIJPEG_MAX_COMPONENTS 4

/* Error codes: numeric values of the reference's JPGERR_* (interface/parameters.hpp:1156-1228). */
#define MIJPEG_OK 0
#define MIJPEG_ERR_INVALID_PARAMETER (-1024)
#define MIJPEG_ERR_UNEXPECTED_EOF (-1025)
#define MIJPEG_ERR_UNEXPECTED_EOB (-1026)
#define MIJPEG_ERR_STREAM_EMPTY (-1027)
#define MIJPEG_ERR_OVERFLOW_PARAMETER (-1028)
#define MIJPEG_ERR_NOT_AVAILABLE (-1029)
#define MIJPEG_ERR_OBJECT_EXISTS (-1030)
#define MIJPEG_ERR_OBJECT_DOESNT_EXIST (-1031)
#define MIJPEG_ERR_MISSING_PARAMETER (-1032)
#define MIJPEG_ERR_BAD_STREAM (-1033)
#define MIJPEG_ERR_OPERATION_UNIMPLEMENTED (-1034)
#define MIJPEG_ERR_PHASE_ERROR (-1035)
#define MIJPEG_ERR_NO_JPG (-1036)
#define MIJPEG_ERR_DOUBLE_MARKER (-1037)
#define MIJPEG_ERR_MALFORMED_STREAM (-1038)
#define MIJPEG_ERR_NOT_IN_PROFILE (-1040)
#define MIJPEG_ERR_THREAD_ABORTED (-1041)
#define MIJPEG_ERR_INVALID_HUFFMAN (-1042)
#define MIJPEG_ERR_OUT_OF_MEMORY (-2048)
#define MIJPEG_ERR_DEVICE (-8191) /* HIP runtime failure (no reference equivalent) */

/* flags for the reconstruct calls */
#define MIJPEG_FLAG_NO_COLOR_TRANSFORM 1u /* JPGTAG_MATRIX_LTRAFO = JPGFLAG_MATRIX_COLORTRANSFORMATION_NONE (CLI -c) */
#define MIJPEG_FLAG_FORCE_GENERIC 2u      /* use the unfused generic kernels even where a fused one exists (testing) */
#define MIJPEG_FLAG_FORCE_SAFE 4u         /* use the 32/64-bit "safe" arithmetic flavour even if the range check passed */
#define MIJPEG_FLAG_NO_UPSAMPLING 16u      /* mijpeg_reconstruct_rect: JPGTAG_DECODER_UPSAMPLE = false -- one component on its own sample grid */
#define MIJPEG_FLAG_DEVICE_OUTPUT 8u      /* mijpeg_reconstruct_rect: dst[] are DEVICE pointers; nothing crosses PCIe */
#define MIJPEG_FLAG_FORCE_DOT2 64u        /* (testing) the 16-bit second pass of the packed 4:2:0 kernel whatever the range check says */
#define MIJPEG_FLAG_SPECULATIVE 32u       /* mijpeg_reconstruct_batch_device with sync = 0 on a SUBMITTED batch: launch the reconstruction
                                             behind the Huffman kernel without waiting for that kernel's report (see there) */

typedef struct mijpeg_decoder mijpeg_decoder;

/* Frame geometry, as JPEG::GetInformation reports it plus the coefficient-plane layout. */
typedef struct mijpeg_info {
  int32_t width, height;     /* JPGTAG_IMAGE_WIDTH / HEIGHT                                     */
  int32_t components;        /* JPGTAG_IMAGE_DEPTH                                              */
  int32_t precision;         /* JPGTAG_IMAGE_PRECISION (8 on this path)                         */
  int32_t hsamp[MIJPEG_MAX_COMPONENTS], vsamp[MIJPEG_MAX_COMPONENTS]; /* SOF Hi, Vi              */
  int32_t subx[MIJPEG_MAX_COMPONENTS], suby[MIJPEG_MAX_COMPONENTS];   /* JPGTAG_IMAGE_SUBX/SUBY  */
  int32_t quant_index[MIJPEG_MAX_COMPONENTS];                         /* SOF Tqi                 */
  int32_t mcus_x, mcus_y;    /* interleaved MCU grid                                            */
  int32_t blocks_w[MIJPEG_MAX_COMPONENTS], blocks_h[MIJPEG_MAX_COMPONENTS]; /* plane size, blocks */
  int32_t restart_interval;  /* DRI                                                             */
  int32_t ycbcr;             /* 1 = L-transformation is YCbCr->RGB (codestream/tables.cpp:2021-2030) */
  int32_t fast_arith;        /* set by decode_coefficients: 1 = every block passed the range check */
  int64_t coef_offset[MIJPEG_MAX_COMPONENTS]; /* start of each component plane, in int16 units   */
  int64_t coef_count;        /* total int16 coefficients of one frame (all planes)              */
  uint16_t quant[4][64];     /* DQT deltas in natural order, index = Tq                          */
  int32_t range_max[MIJPEG_MAX_COMPONENTS]; /* set by decode_coefficients: max over the component's blocks of
                                sum_k |c_k| * q_k (bounds every IDCT output by 4 * range_max) */
  int32_t sample_bytes;      /* bytes per output sample: 1 (precision 8), 2 (precision 12, JPEG XT)          */
  int32_t xt;                /* 1 = JPEG XT profile C stream (three components or one): see mijpeg_xt_params     */
  int32_t is_float;          /* JPGTAG_IMAGE_IS_FLOAT: the 16-bit codes are half-float bit patterns          */
  int32_t progressive;       /* 1 = progressive frame (SOF2): informational, the reconstruction is the same         */
  int32_t coef_wide;         /* 1 = the planes hold int32 coefficients, two int16 slots each (coef_offset[] and coef_count
                                stay in int16 units and count both slots).  Set by mijpeg_decode_coefficients for the
                                streams -- damaged ones -- whose DC prediction or point transform leaves the 16-bit range:
                                the reference keeps LONG coefficients (coding/blockrow.hpp) and so does this frame; it is
                                reconstructed by the unfused kernels with the reference's 32-bit transform          */
  int32_t dnl;               /* 1 = the frame header carried zero lines and the height arrived in a DNL marker behind the first
                                scan.  The reference has set up its buffers by then and the traces stay in the picture
                                (DESIGN.md "DNL frames"): block rows are created MCU row by MCU row without a bound until
                                the marker is seen (control/blockbuffer.cpp:212-265), and the upsamplers never learn the
                                height (control/blockbitmaprequester.cpp:298-322, upsampling/upsamplerbase.cpp:61-75): their
                                line buffers have no bottom edge.  For such frames blocks_h[] counts one MCU row more than
                                mcus_y * vsamp[] -- the row the first scan creates behind the picture when it meets the marker
                                only there -- and rows[] below tells how far the scans really got                        */
  int32_t rows[MIJPEG_MAX_COMPONENTS]; /* set by decode_coefficients for dnl frames: block rows of the component the scans created;
                                a row below that reads as NULL and transforms to sample value 0 (dct/idct.cpp:336-338)    */
} mijpeg_info;

/* JPEG XT (ISO/IEC 18477-7) profile C parameters of the loaded stream, valid when info.xt != 0:
 * what ColorTransformerFactory::InstallIntegerParameters (colortrafo/colortransformerfactory.cpp:300-594)
 * installs into YCbCrTrafo<UWORD,3,Residual|Extended|ClampFlag|Float,...>.  Supported: explicit (TONE box), parametric
 * (CURV box) or identity L tables, parametric or identity Q and R2 tables, standard or free-form (MTRX box) L, R and C
 * transformations, residual codestream = Huffman sequential / progressive 8..12 bit with the fixpoint DCT or the DCT
 * bypass, hidden refinement scans (-R / -rR of the reference encoder: FINE / RFIN boxes, up to four bits each); merging
 * specifications WITHOUT a residual codestream (the L chain alone: no_residual); the lossless / near-lossless files of part 8 that
 * need no integer DCT (-ro, -Q 100: residual scan types FFB1 / FFB2, RCT, no clamping; rct, rbits, clamp below); alpha channels
 * (mijpeg_alpha_channel).
 * Not supported (JPGERR_NOT_IMPLEMENTED): the integer (lifting) DCT of part 8 (-l, -rl), profiles A / B (float tables, pre/post
 * scaling), arithmetic coding. */
typedef struct mijpeg_xt_params {
  mijpeg_info residual;      /* residual codestream: geometry and quantiser tables; its coef_offset[] are offsets
                                into the SAME per-frame coefficient buffer, behind the legacy planes            */
  int32_t ltable[3][4096];   /* L lookup tables per component: 8 + hidden_bits in, 16 bit out (ltable_entries used)  */
  int32_t ltable_entries;    /* 256 << hidden_bits                                                               */
  int32_t hidden_bits;       /* RSPC: low bits of the legacy coefficients that arrived in hidden refinement scans
                                (FINE boxes); the legacy frame reconstructs at precision 8 + hidden_bits         */
  int32_t residual_hidden_bits; /* ... of the residual coefficients (RFIN boxes): precision 12 + residual_hidden_bits */
  int32_t residual_wide;     /* 1: the residual planes hold int32 coefficients (two int16 slots each)           */
  int32_t ltrafo_ycbcr;      /* L transformation: 1 = YCbCr -> RGB, 0 = identity                                */
  int32_t rtrafo_ycbcr;      /* R transformation                                                                */
  int32_t out_max;           /* 2^(8 + extra range bits) - 1: 65535 (half-float codes, 16-bit integers) or 255              */
  int32_t out_shift;         /* (out_max + 1) / 2                                                               */
  int32_t is_float, clamp;   /* output conversion box: cast to float, clamping                                  */
  /* Beyond what the reference's encoder writes by default (all of it decodes like the reference, bit for bit):
   * free-form L / R / C transformations (MTRX boxes, the encoder's -xyz / -cxyz), parametric curves (CURV boxes) as L, Q
   * or R2 tables, a residual DCT bypass (RDCT box).  `general` != 0 when any of these is in use: the reconstruction
   * then runs the unfused kernels with the literal 64-bit merge and table gathers. */
  int32_t general;
  int32_t lmat[9], rmat[9], cmat[9]; /* 13 fractional bits (ColorTrafo::FIX_BITS); the standard matrices when not free-form */
  int32_t rdct_bypass;       /* RDCT box: no residual DCT, samples = coefficient * (delta[63] << 4) + 2^(Pr-1)
                                (control/residualblockhelper.cpp:203-231) */
  int32_t noise_shaping;     /* ... with the 2x2 averaging of that function                                         */
  int32_t qtable_entries;    /* 2^(residual precision + hidden bits + 4)                                             */
  const int32_t *qtable[3];  /* HOST memory, owned by the decoder object: Q table per component, qtable_entries each;
                                NULL = the identity (a shift)                                                        */
  const int32_t *r2table[3]; /* HOST memory: R2 table per component, (out_max + 1) << 4 entries (2^20 at 16 bits of output);
                                NULL = the identity (x + 8) >> 4                                                     */
  int32_t no_residual;       /* 1: nothing is merged (rr = m_lOutDCShift, colortrafo/ycbcrtrafo.cpp:744-746), the legacy picture goes
                                through the L chain alone.  Either the legacy codestream never came to an EOI marker and the
                                reference has not parsed the residual codestream (codestream/image.cpp:1416-1431): the residual
                                planes are zeros and the merge ignores them.  Or the file has a merging specification and NO
                                residual codestream (`jpeg -R n` without `-r` from HDR / 16-bit input; the transformer with R
                                transformation "zero", colortrafo/colortransformerfactory.cpp:262-283): residual.components is 0,
                                the coefficient store ends behind the legacy planes                                            */
  int32_t ltrafo_standard;   /* 1: the L transformation is the STANDARD YCbCr one (not a free-form matrix): the one a request without
                                colour transformation (MIJPEG_FLAG_NO_COLOR_TRANSFORM, the command line's -c) replaces by the
                                identity -- and nothing else of the merge (colortrafo/colortransformerfactory.cpp:231-232)          */
  /* Lossless / near-lossless coding (part 8: the reference encoder's -ro and -Q 100): the residual codestream is of the residual scan
   * type (SOF 0xffb1, no DCT: rdct_bypass), merged through the reversible colour transformation and without clamping */
  int32_t rct;               /* R transformation = RCT (colortrafo/ycbcrtrafo.cpp:752-766); rtrafo_ycbcr is 0 then                     */
  int32_t rbits;             /* fractional bits of the residual path: 4; 1 with the RCT (a precision bit: Q tables of 2^Pr entries); 0 for
                                the identity under the lossless flag (Tables::FractionalColorBitsOf, codestream/tables.cpp:1621-1660)  */
} mijpeg_xt_params;

/* ---- decoder object (one image at a time; one object = one host thread at a time) ---------- */

/* device >= 0: HIP device ordinal, coefficient store is pinned host memory + a device mirror.
 * device  < 0: host-only object (header parsing and entropy decoding, no GPU needed). */
int mijpeg_create(mijpeg_decoder **out, int device);
void mijpeg_destroy(mijpeg_decoder *d);

/* The byte range is borrowed until the next set_input / destroy. */
int mijpeg_set_input(mijpeg_decoder *d, const uint8_t *data, size_t size);

/* SOI .. first SOS: tables, frame header.  Replaces Decoder::ParseHeaderIncremental +
 * Image::StartParseFrame (codestream/decoder.cpp:77, codestream/image.cpp:660). */
int mijpeg_read_header(mijpeg_decoder *d, mijpeg_info *info);

/* Entropy-decode all scans into the planar int16 coefficient store (natural order, quantised):
 * replaces the Scan::ParseMCU loop of JPEG::ReadInternal (interface/jpeg.cpp:300-340) =
 * SequentialScan::ParseMCU/DecodeBlock (codestream/sequentialscan.cpp:381-428, 678-773).
 * threads <= 0: one per host core.  Restart intervals decode in parallel.  With a device, MCU-row
 * bands are hipMemcpyAsync'ed to the GPU while later bands are still being decoded. */
int mijpeg_decode_coefficients(mijpeg_decoder *d, int threads);

/* The same on the GPU: the host only parses the headers and locates the restart markers, the compressed
 * bytes are uploaded (a few MB instead of ~100 MB of coefficients) and one device thread decodes one restart interval.
 * Available for single-scan Huffman sequential 8-bit frames with a restart interval and at least min_intervals
 * restart intervals (<= 0: library default, below which the host decoder is the faster one); otherwise -- and for every
 * stream that turns out to be damaged (irregular restart markers, an error inside an interval, a virtual interval that does
 * not end where the next begins): the reference's behaviour on those is that of its sequential walk, which the host decoder
 * restates -- it returns MIJPEG_ERR_NOT_AVAILABLE and the caller uses mijpeg_decode_coefficients on the same object.
 * After it, mijpeg_coefficients() downloads the planes on first use. */
int mijpeg_decode_coefficients_device(mijpeg_decoder *d, int min_intervals);

/* Batches (SURVEY config 4: many frames of one shape).  n codestreams of identical width, height and
 * sampling, each qualifying for mijpeg_decode_coefficients_device, are parsed on the host in parallel, uploaded
 * and entropy-decoded by ONE kernel launch (every workgroup works on one image; the restart intervals of all images fill
 * the device, which a single image rarely does), into n coefficient stores that mijpeg_reconstruct_batch_device turns
 * into n frames (`frame_stride` bytes apart, interleaved samples, `row_stride` bytes per line) with ONE launch of the
 * reconstruction kernel.  Nothing but the compressed bytes crosses PCIe.  MIJPEG_ERR_NOT_AVAILABLE when the streams do
 * not form such a batch.  The stream bytes are only read during the call.
 * Images may bring quantisation tables of their own (motion JPEG under rate control): the reconstruction launch then
 * reads per-frame tables from device memory (mijpeg_batch.quant_dev below) and mijpeg_get_info reports, per component,
 * the largest delta of any image.
 * Streams: every call of a decoder object works on the object's own non-blocking stream.  Device memory handed to it
 * (dst_device, device bitmaps, pixels to encode) must not have work pending on other streams when the call is made, and
 * is complete when a call returns with sync != 0. */
int mijpeg_decode_batch_device(mijpeg_decoder *d, const uint8_t *const *streams, const size_t *sizes, int n, int min_intervals);
/* The same in two halves, for pipelines that keep several decoder objects in flight from one thread: submit parses the headers,
 * gathers the streams into pinned memory and ENQUEUES upload and Huffman kernel on the object's stream, then returns (the stream
 * bytes are only read during the call); finish waits for them and evaluates what the kernel reported (errors, range check).
 * mijpeg_reconstruct_batch_device and mijpeg_get_info finish a submitted batch themselves.  While one object's batch is on the
 * device the host prepares the next one with another object: see libjpeg_amd/batch.py (BASELINE config 4).  Batches of streams
 * without restart markers get a fixed, generous number of device-walk rounds (rounds in which nothing changes cost
 * microseconds); finish answers MIJPEG_ERR_NOT_AVAILABLE in the unlikely case that the walk had not settled by then, and
 * mijpeg_decode_batch_device, which looks at the walk between its rounds, decodes such a batch. */
int mijpeg_submit_batch_device(mijpeg_decoder *d, const uint8_t *const *streams, const size_t *sizes, int n, int min_intervals);
int mijpeg_finish_batch_device(mijpeg_decoder *d);
/* Pipelines that never block on the device: mijpeg_reconstruct_batch_device(.., flags | MIJPEG_FLAG_SPECULATIVE, sync = 0) on a batch
 * that was only SUBMITTED launches the reconstruction right behind the Huffman kernel, on the range check (the kernel selection's
 * input, which the Huffman kernel reports) of the last finished batch of the same frame shape and tables, rounded up to the next
 * gate of the selection -- instead of waiting for this batch's.  mijpeg_finish_batch_device then VALIDATES: it waits, evaluates what
 * the Huffman kernel reported (damaged streams: the usual MIJPEG_ERR_NOT_AVAILABLE) and, should the real range check lie beyond the
 * assumed one, runs the reconstruction again with the kernel it selects (synchronously, into the same destination).  Until it has
 * returned MIJPEG_OK the destination's pixels do not count; mijpeg_synchronize and the next mijpeg_submit_batch_device validate as
 * well (and report the batch's error).  Without a finished batch of this shape to go by, with per-image tables, or for streams
 * without restart markers the call behaves as without the flag.  mijpeg_batch_speculation: diagnostics -- returns 1 when the last
 * validation had to reconstruct again; *launched / *redone count the object's speculative launches and the ones redone. */
int mijpeg_batch_speculation(mijpeg_decoder *d, int64_t *launched, int64_t *redone);
/* BASELINE config 4 as ONE call (round 6; libjpeg_amd/csrc/batch_pipeline.cpp -- until then a Python loop): `decoder_objects`
 * decoder objects on `device`, driven round-robin by the calling thread over chunks of `chunk_frames` streams (ramp != 0:
 * smaller chunks at both ends of the batch): submit of chunk i + 1 (host: parse, marker search, gather) while chunk i's upload,
 * Huffman kernel and fused kernel run.  run: n streams of one shape, bytes in host memory -> frame i at dst_device + i *
 * frame_stride (device memory, `row_stride` bytes per line); returns when every frame is there.  download_host != NULL: the
 * frames also travel to that (pinned) host buffer, same strides, each chunk as soon as its reconstruction is through, on a
 * stream of the pipeline's own; run returns when the last one has arrived.  A chunk the device path declines
 * (MIJPEG_ERR_NOT_AVAILABLE) is decoded by mijpeg_decode_batch_device, failing that stream by stream on the host; any other
 * error ends the run with its code (mijpeg_batch_pipeline_last_error).  stats: chunks of the last run, how many fell back, host
 * milliseconds of every submit.  schedule: the chunk boundaries run would use.  decoder: object k (diagnostics: mijpeg_get_info,
 * mijpeg_last_timing of its last chunk). */
typedef struct mijpeg_batch_pipeline mijpeg_batch_pipeline;
int mijpeg_batch_pipeline_create(mijpeg_batch_pipeline **out, int device, int chunk_frames, int decoder_objects, int ramp);
void mijpeg_batch_pipeline_destroy(mijpeg_batch_pipeline *p);
int mijpeg_batch_pipeline_run(mijpeg_batch_pipeline *p, const uint8_t *const *streams, const size_t *sizes, int n, void *dst_device,
                              int64_t frame_stride, int64_t row_stride, void *download_host);
int mijpeg_batch_pipeline_schedule(int n, int chunk_frames, int ramp, int32_t *first, int32_t *end, int capacity);
int mijpeg_batch_pipeline_stats(mijpeg_batch_pipeline *p, int32_t *chunks, int32_t *fallbacks, float *submit_ms, int capacity);
int mijpeg_batch_pipeline_last_error(mijpeg_batch_pipeline *p, const char **message);
/* on = 1 / 0: launch the reconstructions speculatively (MIJPEG_FLAG_SPECULATIVE above; off by default), on < 0: leave as is.
 * Returns how many chunks of the last run were reconstructed again by their validation. */
int mijpeg_batch_pipeline_speculation(mijpeg_batch_pipeline *p, int on);
mijpeg_decoder *mijpeg_batch_pipeline_decoder(mijpeg_batch_pipeline *p, int k);
/* Capacity planning / diagnostics: the HOST half of mijpeg_submit_batch_device alone -- header parse, restart marker search
 * and the copy of the entropy coded data without its byte stuffing into the staging area, one stream per pool worker -- with
 * no device involved (works on host-only objects).  This is what a rank's cores do per chunk of a batch; `bench.py
 * --emulate-world N` runs it in neighbour processes to load the host the way the other ranks of a node would. */
int mijpeg_prepare_batch_host(mijpeg_decoder *d, const uint8_t *const *streams, const size_t *sizes, int n);
/* Wait for everything the object has enqueued on its stream (e.g. a reconstruction launched with sync = 0). */
int mijpeg_synchronize(mijpeg_decoder *d);
/* ... or let a stream of the CLIENT's wait for it instead of the host: everything the object has enqueued so far (uploads, entropy
 * kernels, a reconstruction launched with sync = 0) happens before whatever the client enqueues on `client_stream` (a hipStream_t;
 * NULL = the null stream) after this call.  The host does not block.  What config 4's other half uses: the download of a chunk's
 * pixels waits for that chunk's reconstruction kernel and for nothing else, so PCIe carries pixels down while the next chunks'
 * bytes go up (libjpeg_amd/batch.py, `download_to`). */
int mijpeg_stream_wait(mijpeg_decoder *d, void *client_stream);
int mijpeg_reconstruct_batch_device(mijpeg_decoder *d, void *dst_device, int64_t frame_stride, int64_t row_stride, uint32_t flags,
                                    int sync);

/* Current frame information (after mijpeg_decode_coefficients it includes fast_arith). */
int mijpeg_get_info(mijpeg_decoder *d, mijpeg_info *info);

/* JPEG XT parameters of the loaded stream (MIJPEG_ERR_OBJECT_DOESNT_EXIST if it is a plain JPEG). */
int mijpeg_get_xt_params(mijpeg_decoder *d, mijpeg_xt_params *xt);

/* Host view of a decoded component plane: blocks_h x blocks_w x 64 int16 (NULL for a frame with info.coef_wide). */
const int16_t *mijpeg_coefficients(mijpeg_decoder *d, int component);
/* ... of a frame with info.coef_wide: blocks_h x blocks_w x 64 int32 (NULL for every other frame). */
const int32_t *mijpeg_coefficients32(mijpeg_decoder *d, int component);

/* Device pointer of the uploaded frame (coef_count int16) or NULL. */
const int16_t *mijpeg_device_coefficients(mijpeg_decoder *d);

/* Reconstruct the whole frame on the GPU into DEVICE memory: interleaved 8-bit samples,
 * `components * sample_bytes` bytes per pixel, `row_stride` bytes per line.  Asynchronous on the decoder's stream
 * unless `sync` is non-zero. */
int mijpeg_reconstruct_device(mijpeg_decoder *d, void *dst_device, int64_t row_stride, uint32_t flags,
                              int sync);

/* Reconstruct the whole frame into HOST memory (interleaved samples, `row_stride` bytes per line) with the
 * device-to-host copy landing directly in `dst_host`: no staging copy when the memory is pinned
 * (mijpeg_host_alloc, hipHostMalloc, hipHostRegister); pageable memory works too, only slower. Synchronous. */
int mijpeg_reconstruct_host(mijpeg_decoder *d, void *dst_host, int64_t row_stride, uint32_t flags);

/* Pinned host memory for frame buffers (hipHostMalloc / hipHostFree). */
void *mijpeg_host_alloc(size_t bytes);
void mijpeg_host_free(void *p);

/* Reconstruct a rectangle into HOST memory described like the reference's ImageBitMap
 * (interface/imagebitmap.hpp): for component c, dst[c] is the address of canvas pixel (0,0),
 * bytes_per_pixel[c] / bytes_per_row[c] the strides.  Rectangle and component range are inclusive,
 * as in JPGTAG_DECODER_MINX..MAXY / MINCOMPONENT..MAXCOMPONENT (codestream/rectanglerequest.cpp:92-152).
 * With MIJPEG_FLAG_NO_UPSAMPLING (min_comp == max_comp required) the component is delivered at its own resolution
 * without colour transformation; the rectangle is still given on the canvas and shrinks by the subsampling factors
 * (control/bitmapctrl.cpp:273-294), dst[c] addresses the component's sample (0,0).  JPEG XT frames return
 * MIJPEG_ERR_OPERATION_UNIMPLEMENTED for it: the reference merges the one requested component with whatever its scratch
 * buffers hold for the residuals of the OTHER two -- heap memory it never initialised during the first component's pass
 * (control/blockbitmaprequester.cpp:379-388, 1054-1061; `MALLOC_PERTURB_=165 jpeg -U xt.jpg out.pgx` writes other planes
 * than `jpeg -U xt.jpg out.pgx`), so there is no output to equal.
 * With MIJPEG_FLAG_DEVICE_OUTPUT the bitmaps live in device memory of the decoder's GPU and the pixels never leave
 * HBM ("device-side output"); the call returns when they are written. */
int mijpeg_reconstruct_rect(mijpeg_decoder *d, int32_t min_x, int32_t min_y, int32_t max_x, int32_t max_y,
                            int32_t min_comp, int32_t max_comp, uint32_t flags,
                            void *const dst[MIJPEG_MAX_COMPONENTS],
                            const int32_t bytes_per_pixel[MIJPEG_MAX_COMPONENTS],
                            const int32_t bytes_per_row[MIJPEG_MAX_COMPONENTS]);

/* JPEG::DisplayRectangle AS THE REFERENCE RUNS IT: one call of a sequence.  mijpeg_reconstruct_rect above answers every
 * request from the plain picture; the reference does not -- BlockBitmapRequester::ReconstructRegion
 * (control/blockbitmaprequester.cpp:1013-1272) walks row cursors that never rewind and upsampler line buffers
 * (upsampling/upsamplerbase.cpp:138-327) that persist between calls, so a request shows what the calls before it left:
 * on the upsampling path every call moves the cursor of every unsubsampled component, requested or not (:1214-1223 -- the
 * reference's own PGX loop, cmd/reconstruct.cpp:272-303, therefore writes planes of zeros for the later unsubsampled
 * components of a two- or four-component frame with mixed sampling); components outside the requested range enter the colour
 * transformation as zeros; the colour transformer of the first request stays (colortrafo/colortransformerfactory.cpp:220-221);
 * rectangles whose corner is off the 8-pixel grid see subsampled components displaced (upsampling/upsampler.cpp:85-86 against
 * colortrafo/ycbcrtrafo.cpp:683-686); stripes that are skipped or repeated read the rows the cursors stand at.  This entry
 * point keeps that state per decoder object (reset by every decode) and reproduces all of it; top-down requests of whole
 * component sets -- every sane client -- are recognised as the plain picture and served from the cached frame.
 * bitmaps[c] describes what the bitmap hook returned for component c (requested components only are read): address of canvas
 * pixel (0,0), strides in bytes, BIO_WIDTH / BIO_HEIGHT (the height bounds the block rows that are reconstructed, :1229-1244;
 * blocks that start outside the extent are not written, interface/imagebitmap.cpp:58-129).  data == NULL: nothing is
 * written for that component but the state advances.  flags: MIJPEG_FLAG_NO_UPSAMPLING, _NO_COLOR_TRANSFORM, _DEVICE_OUTPUT.
 * JPEG XT frames: the residual image has cursors and upsamplers of its own beside the legacy image's (:228-232, 356-372,
 * 1118-1146, 1197-1222) and both follow every request -- reproduced for what the reference's command line asks for (all three
 * components, upsampling and colour transformation on) with any order and size of rectangles; a request that walks the
 * residual cursor of a component without upsampler behind its last row makes the reference dereference a NULL row
 * (:1057-1058, :1201-1202): MIJPEG_ERR_OBJECT_DOESNT_EXIST here.  Component subsets of an XT frame (they merge with what a
 * scratch buffer holds from the block before) are served as the plain picture, XT requests without colour transformation as
 * the plain picture with the identity in place of the standard YCbCr L transformation (what the reference's transformer
 * does, mijpeg_xt_params.ltrafo_standard), XT requests without upsampling fail (see mijpeg_reconstruct_rect).  Not modelled either: rectangles narrower than the frame that move sideways between calls read line-buffer
 * memory the reference never initialised. */
typedef struct mijpeg_bitmap {
  void *data;
  int32_t bytes_per_pixel, bytes_per_row;
  uint32_t width, height;
} mijpeg_bitmap;
int mijpeg_display_rect(mijpeg_decoder *d, int32_t min_x, int32_t min_y, int32_t max_x, int32_t max_y, int32_t min_comp, int32_t max_comp,
                        uint32_t flags, const mijpeg_bitmap bitmaps[MIJPEG_MAX_COMPONENTS]);
/* Diagnostics, no device needed (mijpeg_read_header is enough): advance the request state exactly like mijpeg_display_rect
 * would and report the plan instead of pixels -- out[0..7] = nothing shown, plain picture, YCbCr transformer, view component,
 * region min_x, min_y, max_x, max_y; then per component: cursor after the call, first / last block row with a row map, upsampler
 * window start / end (lines), number of mapped block rows that are zeros. */
int mijpeg_display_plan(mijpeg_decoder *d, int32_t min_x, int32_t min_y, int32_t max_x, int32_t max_y, int32_t min_comp, int32_t max_comp,
                        uint32_t flags, const uint32_t bm_height[MIJPEG_MAX_COMPONENTS], int32_t out[8 + 6 * MIJPEG_MAX_COMPONENTS]);
/* Diagnostics: coefficient row the cursor of `component` stands at after the mijpeg_display_rect calls so far (JPEG XT:
 * 4 + c = component c of the residual image, which has cursors of its own, control/blockbitmaprequester.cpp:228-232). */
int mijpeg_display_cursor(mijpeg_decoder *d, int component);

/* The scans of the decoded frame in codestream order: first_byte[k] = offset (in the input of mijpeg_set_input) of the first
 * entropy coded byte of scan k, i.e. where the reference stands when JPEG::Read returns with JPGFLAG_DECODER_STOP_SCAN
 * (interface/jpeg.cpp:310-353), end_byte[k] = offset of the marker that follows the scan's data.  JPEG XT scans that live in
 * boxes come last: first_byte = the marker behind the codestream's last scan (where the reference's input stands while it
 * parses them from memory), end_byte = 0 -- the legacy frame's refinement scans, the residual codestream's scans, then the
 * alpha channel's codestreams in the same order (Image::ParseAlphaChannel, codestream/image.cpp:1337-1404).  A frame that
 * was decoded by the sequential walk (a height from a DNL marker, the residual scan types of part 8, damaged streams) is
 * listed by the scans that walk met.  Either array may be NULL.  Returns the number of scans (also when capacity is
 * smaller) or a negative error. */
int mijpeg_scan_offsets(mijpeg_decoder *d, uint64_t *first_byte, uint64_t *end_byte, int capacity);
/* ... and their MCU grids, same order: mcus_y[k] rows of mcus_x[k] MCUs (a single-component scan walks its component's own blocks,
 * codestream/sequentialscan.cpp:396-397).  What JPEG::Read with JPGFLAG_DECODER_STOP_ROW / _MCU counts its returns by: one at the
 * start of every MCU row, one behind every MCU of a row but the last (interface/jpeg.cpp:326-350).  (The scan that brings a DNL
 * marker: as many rows as the reference starts to find it.)  Returns the number of scans. */
int mijpeg_scan_grids(mijpeg_decoder *d, int32_t *mcus_x, int32_t *mcus_y, int capacity);

/* JPEG XT alpha channel (the reference: Image::ParseAlphaChannel, codestream/image.cpp:1337-1404; JPEG::GetInformation's
 * JPGTAG_ALPHA_MODE / JPGTAG_ALPHA_TAGLIST / JPGTAG_ALPHA_MATTE, interface/jpeg.cpp:870-945; JPEG::DisplayRectangle with
 * JPGTAG_BIH_ALPHAHOOK and JPGTAG_DECODER_INCLUDE_ALPHA, codestream/image.cpp:1087-1123, cmd/reconstruct.cpp:154-217, 268-319).
 * The alpha channel is an image of its own -- one component, its own tables, optionally its own residual codestream and hidden
 * refinement scans, its own merging specification -- that the reference decodes inside JPEG::Read behind the picture's
 * codestreams: mijpeg_decode_coefficients (host or device) decodes it as well, and what is wrong with it fails that call like it
 * fails JPEG::Read.  The decoder does not composite: the plane is handed out beside the picture.
 * mijpeg_alpha_channel: the decoder object of the alpha image, owned by `d` and valid until `d` decodes again or is destroyed; every
 * call that works on a decoded object works on it (mijpeg_get_info, mijpeg_get_xt_params, mijpeg_reconstruct_rect,
 * mijpeg_display_rect with cursors of its own, mijpeg_coefficients).  NULL (and MIJPEG_ERR_OBJECT_DOESNT_EXIST on `d`) when the
 * decoded file has none -- no complete ALFA box, or a legacy codestream that never came to its EOI.
 * mijpeg_alpha_info: *mode = the compositing method of the AMUL box (0 opaque, 1 regular, 2 premultiplied, 3 matte removal;
 * -1: the alpha merging specification has no such box and JPEG::GetInformation reports no alpha channel), matte = its colour. */
mijpeg_decoder *mijpeg_alpha_channel(mijpeg_decoder *d);
/* 1 when mijpeg_alpha_channel(d) would hand out a decoder, 0 otherwise; never records an error (JPEG::GetInformation asks on every file,
 * interface/jpeg.cpp:919-951, and must not leave "no alpha channel" behind as the object's last error). */
int mijpeg_has_alpha(mijpeg_decoder *d);
int mijpeg_alpha_info(mijpeg_decoder *d, int32_t *mode, int32_t matte[3]);

/* Error of the last failing call on this object (JPEG::LastError). Returns the code, 0 if none. */
int mijpeg_last_error(mijpeg_decoder *d, const char **message);

/* JPEG::LastWarning (interface/jpeg.cpp:970-979): 0, or the code of what the reference warns about at the loaded stream --
 * MIJPEG_ERR_PHASE_ERROR: the LCHK checksum box of a JPEG XT file does not fit its legacy codestream (interface/jpeg.cpp:222-238;
 * computed when first asked for), MIJPEG_ERR_MALFORMED_STREAM: a damaged codestream that was decoded with resynchronisation. */
int mijpeg_last_warning(mijpeg_decoder *d, const char **message);

/* Diagnostics: the entropy coded data of the first scan of the parsed stream (mijpeg_read_header is not enough:
 * mijpeg_decode_coefficients or a device decode must have parsed it) the way the device decoder receives it -- without the
 * byte stuffing and without the markers, restart interval k at begin[k] (n_begin entries are filled at most), copied in
 * pieces of at most piece_bytes source bytes like the parallel gather does (piece_bytes = 1: the stream is parsed again and the
 * marker search writes the copy while it walks the segment, as a batch's workers do; capacity >= stream size + 64 then).  Returns the number of bytes (also when dst is
 * NULL or too small: nothing is copied then), or a negative error code.  *n_intervals (may be NULL): restart intervals. */
int64_t mijpeg_unstuffed_scan(mijpeg_decoder *d, uint8_t *dst, size_t capacity, uint32_t *begin, size_t n_begin, size_t piece_bytes,
                              int32_t *n_intervals);

/* Diagnostics: number of scans without restart markers that the host decoder decoded in parallel (self-synchronising
 * speculative decoding) since the library was loaded; *pieces (may be NULL) = ranges they were stitched from. */
int64_t mijpeg_speculative_scans(int64_t *pieces);

/* Diagnostics: rounds the on-device self-synchronising walk needed in the last device entropy decode of streams
 * without restart markers (0: no such walk took place, e.g. the streams had restart markers). */
int mijpeg_device_walk_rounds(mijpeg_decoder *d);

/* Seconds spent in the phases of the last decode (huffman, h2d, kernel, d2h) -- diagnostics. */
int mijpeg_last_timing(mijpeg_decoder *d, double out_seconds[4]);

/* ---- stateless device entry points (inputs and outputs resident in HBM) -------------------- */

/* Describes a batch of equally shaped frames whose coefficient planes are already on the device. */
typedef struct mijpeg_batch {
  mijpeg_info info;            /* geometry + quantiser tables (shared by the batch unless quant_dev: then the
                                  element-wise maxima over the frames, which the kernel selection looks at,
                                  and range_max / fast_arith of the most demanding frame) */
  const int16_t *coef_dev;     /* frame f starts at coef_dev + f * coef_frame_stride (int16 units) */
  int64_t coef_frame_stride;
  const uint16_t *quant_dev;   /* optional: per-frame tables [frames][4][64] u16 on the device: deltas of
                                  COMPONENT c of frame f, natural order; plain JPEG only.  They are expanded
                                  into the workspace (see mijpeg_workspace_bytes) by a small kernel in front */
  uint8_t *out_dev;            /* frame f pixels at out_dev + f * out_frame_stride (bytes)          */
  int64_t out_frame_stride;
  int64_t out_row_stride;      /* bytes per line                                                  */
  int32_t frames;
  uint32_t flags;
  void *workspace;             /* device scratch for the unfused path, see mijpeg_workspace_bytes     */
  size_t workspace_bytes;
  const mijpeg_xt_params *xt;  /* required when info.xt != 0 (host memory)                            */
} mijpeg_batch;

/* Device scratch the batch needs (0 for the fused kernels of plain JPEG; the L tables for the fused JPEG XT kernel;
 * tables + int32 sample planes for the generic kernels; plus frames x 1 KiB of expanded tables with quant_dev). */
size_t mijpeg_workspace_bytes(const mijpeg_batch *batch);

/* dequant + IDCT + upsample + colour transform + interleaved store for `frames` frames in one
 * launch on `stream` (hipStream_t).  Asynchronous.  This is what bench.py times. */
int mijpeg_launch_reconstruct(const mijpeg_batch *batch, void *stream);

/* Name of the kernel the batch would run ("fused420", "generic", ...) -- for profiling scripts. */
const char *mijpeg_kernel_name(const mijpeg_batch *batch);

/* ---- encoder direction of the block pipeline -----------------------------------------------------
 * What BlockBitmapRequester::PullSourceData / AdvanceQRows (control/blockbitmaprequester.cpp:505-576, 708-846) do
 * per block in front of the entropy coder: forward L transformation (colortrafo/ycbcrtrafo.cpp:85-242), box
 * downsampling (upsampling/downsampler.cpp:70-139), forward DCT + quantisation (dct/idct.cpp:114-222), for frames
 * resident in HBM.  Entropy coding and marker writing are not part of it. */

/* Completes *info from width, height, components (1 or 3), precision (8), hsamp[], vsamp[], quant_index[] and quant[][]:
 * MCU grid, subsampling factors, plane sizes, coef_offset[] and coef_count, as a frame header with these values
 * would produce (marker/frame.cpp, marker/component.cpp).  Returns MIJPEG_OK or MIJPEG_ERR_INVALID_PARAMETER. */
int mijpeg_frame_layout(mijpeg_info *info);

typedef struct mijpeg_forward_batch {
  mijpeg_info info;            /* completed by mijpeg_frame_layout; ycbcr = 1: RGB in, YCbCr coded; 0: identity   */
  const uint8_t *pixels_dev;   /* interleaved 8-bit samples, `components` per pixel; frame f at + f * pixel_frame_stride */
  int64_t pixel_frame_stride;  /* bytes                                                                            */
  int64_t pixel_row_stride;    /* bytes per line                                                                   */
  int16_t *coef_dev;           /* out: quantised coefficients in the layout the decoder reads (natural order inside a
                                  block, blocks row-major, planes at info.coef_offset[]); MCU padding blocks are zero */
  int64_t coef_frame_stride;   /* int16 units                                                                      */
  int32_t frames;
  uint32_t flags;              /* 0                                                                                */
} mijpeg_forward_batch;

/* One launch on `stream` (hipStream_t) for all frames of the batch.  Asynchronous. */
int mijpeg_launch_forward(const mijpeg_forward_batch *batch, void *stream);

/* Entropy coder + stream writer for what mijpeg_launch_forward produced (after a device-to-host copy): quantised
 * coefficient planes in HOST memory, the layout of *info, -> a baseline (SOF0) JPEG stream with one interleaved
 * Huffman-sequential scan (codestream/sequentialscan.cpp:430-676 WriteMCU / EncodeBlock; segment syntax: marker/ directory).
 * restart_interval: MCUs per restart interval, 0 = none (the intervals are coded in parallel on `threads` threads,
 * <= 0: default); optimize != 0: Huffman tables optimised for the picture (Annex K.2) instead of the Annex K.3 tables.
 * MCU padding blocks are coded as "same DC, no AC".  *stream is malloc'ed: release it with mijpeg_free. */
int mijpeg_encode_coefficients(const mijpeg_info *info, const int16_t *coef, int restart_interval, int optimize, int threads,
                               uint8_t **stream, size_t *size);
void mijpeg_free(void *p);

/* The quantiser tables the reference encoder derives from `-q quality` with its default (Annex K) matrices:
 * Quantization::InitDefaultTables, marker/quantization.cpp:275-466, for 8-bit frames -- natural order. */
void mijpeg_quality_tables(int quality, uint16_t luma[64], uint16_t chroma[64]);

/* Whole encoder-direction pipeline for one picture in HOST memory on the decoder object's device: upload, forward kernel,
 * download of the coefficients, entropy coding (`jpeg -bl -q quality -s ... -z restart_interval [-h] in.ppm out.jpg` of the
 * reference CLI, cmd/encodec.cpp; here: Annex K.3 or optimised Huffman tables, one interleaved baseline scan).
 * pixels: interleaved 8-bit, `components` (1 or 3) per pixel, row_stride bytes per line; hsamp/vsamp: sampling factors
 * per component (NULL: 1x1 everywhere); RGB input is coded as YCbCr.  *stream is malloc'ed (mijpeg_free). */
int mijpeg_encode_image(mijpeg_decoder *d, const uint8_t *pixels, int32_t width, int32_t height, int32_t components, int64_t row_stride,
                        int quality, const int32_t *hsamp, const int32_t *vsamp, int restart_interval, int optimize,
                        uint8_t **stream, size_t *size);
/* The same with flags.  By default the entropy coder runs on the device too (hencode.hip: the coefficients never leave
 * HBM, only the finished stream comes down); MIJPEG_ENCODE_HOST_CODER downloads the coefficients and codes them with
 * mijpeg_encode_coefficients.  Both write the same bytes. */
#define MIJPEG_ENCODE_HOST_CODER 1u
int mijpeg_encode_image_ex(mijpeg_decoder *d, const uint8_t *pixels, int32_t width, int32_t height, int32_t components, int64_t row_stride,
                           int quality, const int32_t *hsamp, const int32_t *vsamp, int restart_interval, int optimize, uint32_t flags,
                           uint8_t **stream, size_t *size);

/* Frames that are already in device memory (a renderer's or a video pipeline's output): forward kernels for the whole batch in
 * one launch, then the device entropy coder frame by frame; only the finished streams cross PCIe.  batch->coef_dev is
 * scratch for the coefficient planes (frames * coef_frame_stride int16).  streams[f] is malloc'ed (mijpeg_free). */
int mijpeg_encode_batch_device(mijpeg_decoder *d, const mijpeg_forward_batch *batch, int restart_interval, int optimize,
                               uint8_t **streams, size_t *sizes);

/* Worker threads mijpeg_decode_coefficients uses for threads <= 0 (MIJPEG_THREADS overrides; default min(cores, 64)). */
int mijpeg_default_threads(void);

/* The large buffers of destroyed decoder objects (pinned coefficient store and frame, their device mirrors) wait in a process-wide
 * cache for the next object on the same device -- a client that constructs a JPEG object per picture would otherwise pin and
 * unpin ~200 MB per 8K picture.  At most four buffers per kind and MIJPEG_BUFFER_CACHE_MB megabytes in all (environment,
 * default 2048, 0 = keep nothing) are kept; a buffer enters the cache only after the device has gone idle, so kernels a client
 * launched on its own streams against mijpeg_device_coefficients() / a mijpeg_batch never race with the next owner.
 * mijpeg_trim_cache() hands everything the cache holds back to the runtime and returns the number of bytes freed. */
size_t mijpeg_trim_cache(void);

/* Library / build identification. */
const char *mijpeg_version(void);

#ifdef __cplusplus
}
#endif
#endif
