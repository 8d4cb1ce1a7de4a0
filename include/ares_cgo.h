/* ares_cgo.h — the result handle every entry point of the drop-in boundary returns.
 *
 * Replaces: reference cgoutils/utils.h:20-23 (CGoCallResHandle).
 * Contract (reference cgoutils/utils.go:26-34): success => pStrErr == NULL and `res`
 * carries an integer/pointer result cast to void*; failure => pStrErr is a malloc-family
 * heap string that the Go caller reads, C.free()s and panics on.  16 bytes, returned by
 * value in RAX:RDX under the SysV x86-64 ABI.
 */
#ifndef ARES_CGO_H_
#define ARES_CGO_H_

typedef struct {
  void *res;
  const char *pStrErr;
} CGoCallResHandle;

#endif /* ARES_CGO_H_ */
