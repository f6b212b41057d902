/* peload.h -- TEST INFRASTRUCTURE ONLY: in-process mapper for one Win64 DLL (see peload.c). */
#ifndef SFE_ORACLE_PELOAD_H
#define SFE_ORACLE_PELOAD_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pe_image pe_image;
/* returns the address an import slot of `dll`!`name` is bound to, NULL when there is none */
typedef void *(*pe_resolver)(const char *dll, const char *name);

pe_image *pe_load(const char *path, pe_resolver resolve);   /* map, relocate, bind imports */
int pe_run_entry(pe_image *im);                             /* TLS callbacks + entry(DLL_PROCESS_ATTACH) */
void *pe_export(pe_image *im, const char *name);            /* an ms_abi function or data address */
void pe_enter_thread(void);                                 /* give the calling thread its TEB (idempotent) */
void pe_image_range(void **base, unsigned long *size);      /* for VirtualQuery */
const char *pe_last_error(void);

void *win_stub_resolve(const char *dll, const char *name);  /* win_stubs.c */

#ifdef __cplusplus
}
#endif
#endif
