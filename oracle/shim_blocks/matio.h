/* TEST INFRASTRUCTURE ONLY: matio is absent; the dump path of the blocks compiles against these names and
 * Mat_CreateVer returns nullptr (the blocks then print "Unable to create ..." and carry on -- dump is never enabled
 * by the pin tests). */
#ifndef ORACLE_SHIM_MATIO_H
#define ORACLE_SHIM_MATIO_H
#include <cstddef>
#include <cstdint>
typedef struct mat_t mat_t;
typedef struct matvar_t matvar_t;
enum matio_classes { MAT_C_INT32, MAT_C_UINT32, MAT_C_UINT8, MAT_C_INT64, MAT_C_UINT64, MAT_C_SINGLE, MAT_C_DOUBLE, MAT_C_CHAR };
enum matio_types { MAT_T_INT32, MAT_T_UINT32, MAT_T_UINT8, MAT_T_INT64, MAT_T_UINT64, MAT_T_SINGLE, MAT_T_DOUBLE, MAT_T_UTF8 };
enum mat_ft { MAT_FT_MAT73 };
enum matio_compression { MAT_COMPRESSION_NONE, MAT_COMPRESSION_ZLIB };
static inline mat_t* Mat_CreateVer(const char*, const char*, mat_ft) { return nullptr; }
static inline int Mat_Close(mat_t*) { return 0; }
static inline matvar_t* Mat_VarCreate(const char*, matio_classes, matio_types, int, size_t*, const void*, int) { return nullptr; }
static inline int Mat_VarWrite(mat_t*, matvar_t*, matio_compression) { return 0; }
static inline void Mat_VarFree(matvar_t*) {}
#endif
