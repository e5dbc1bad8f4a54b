/* pixo_png_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see pixo_png_oracle.c). */
#ifndef PIXO_PNG_ORACLE_H
#define PIXO_PNG_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* filter type bytes, png/filter.rs:43-47 */
enum { PO_F_NONE = 0, PO_F_SUB = 1, PO_F_UP = 2, PO_F_AVG = 3, PO_F_PAETH = 4 };
/* FilterStrategy in declaration order, png/mod.rs:345-364 */
enum { PO_S_NONE = 0, PO_S_SUB, PO_S_UP, PO_S_AVERAGE, PO_S_PAETH, PO_S_MINSUM, PO_S_ADAPTIVE, PO_S_ADAPTIVE_FAST, PO_S_BIGRAMS };
uint32_t po_adler32(const uint8_t *data, size_t n);
/* out: height * (width*bpp + 1) bytes.  Returns 0, or -1 for bad arguments. */
int po_png_filter(const uint8_t *data, uint32_t width, uint32_t height, uint32_t bpp, int strategy, int stateful_fast,
                  uint8_t *out, uint32_t *adler);
#ifdef __cplusplus
}
#endif
#endif
