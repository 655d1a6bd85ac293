// Micro-benchmark: issue cost (cycles per instruction, per SIMD) of the instruction kinds a fused dequantize-GEMM main loop is
// made of, on gfx950: alone, with 1 / 2 / 4 wavefronts per SIMD running the same stream, and mixed (matrix + VALU + LDS in one
// wavefront). Every body is one inline-asm block of independent instructions on fixed registers (no compiler scheduling, no
// dependent chains unless the body says so), bracketed by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define CLOB_V "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"

// registers: v10..v17 sources; v20..v27 VALU destinations; v30..v61 LDS destinations; v64..v191 accumulators (8 x 16)
// MFMA operands: A = v[10:13], B = v[14:17]
#define M32(acc) "v_mfma_f32_32x32x16_bf16 v[" acc "], v[10:13], v[14:17], v[" acc "]\n"
#define M16(acc) "v_mfma_f32_16x16x32_bf16 v[" acc "], v[10:13], v[14:17], v[" acc "]\n"
#define VMUL(d) "v_mul_f32 v" d ", v10, v11\n"
#define VFMA(d) "v_fma_f32 v" d ", v10, v11, v12\n"
#define VPKMUL(d) "v_pk_mul_f32 v[" d "], v[10:11], v[12:13]\n"
#define VPKFMA(d) "v_pk_fma_f32 v[" d "], v[10:11], v[12:13], v[14:15]\n"
#define VCVT(d) "v_cvt_pk_bf16_f32 v" d ", v10, v11\n"
#define VPERM(d) "v_perm_b32 v" d ", v10, v11, v12\n"
#define VANDOR(d) "v_and_or_b32 v" d ", v10, v11, v12\n"
#define DSR64(d, off) "ds_read_b64 v[" d "], %0 offset:" off "\n"
#define DSR32(d, off) "ds_read_b32 v" d ", %0 offset:" off "\n"
#define DSR128(d, off) "ds_read_b128 v[" d "], %0 offset:" off "\n"
#define DSW128(off) "ds_write_b128 %0, v[10:13] offset:" off "\n"
#define DSW64(off) "ds_write_b64 %0, v[10:11] offset:" off "\n"
#define WAITL "s_waitcnt lgkmcnt(0)\n"

struct Body { const char* name; int count; }; // count = instructions of the kind named per body execution

enum {
    B_IDLE = 0, B_VMUL, B_VFMA, B_VPKMUL, B_VPKFMA, B_VCVT, B_VPERM, B_VANDOR,
    B_DSR64, B_DSR32, B_DSR128, B_DSW128, B_DSW64,
    B_M32_1, B_M32_2, B_M32_4, B_M16_1, B_M16_2, B_M16_4, B_M16_8,
    B_M32_4_V2, B_M32_4_V4, B_M32_4_V6, B_M32_4_V8, B_M32_4_V6_L1, B_M32_4_V6_L2, B_M32_2_V6_L2, B_M32_2_V6, B_M32_2_V2, B_M32_1_V2, B_M32_2_L1, B_M32_2_L2, B_M32_4_L2, B_M32_2x2_V6_L2, B_M16_2_V2, B_M16_4_V3_L1, B_VMUL16, B_VMULSAME, B_WAR0, B_WAR1, B_WAR2, B_WAR3, B_WAR4,
    B_M16_4_V2, B_M16_4_V4, B_M16_8_V3_L1, B_DECODE_A, B_DECODE_D, B_COUNT
};

template <int B> __device__ __forceinline__ void body(uint32_t lds_addr) {
#define R8(x) x x x x x x x x
#define R4(x) x x x x
    if constexpr (B == B_VMUL) asm volatile(R8(VMUL("20") VMUL("21") VMUL("22") VMUL("23") VMUL("24") VMUL("25") VMUL("26") VMUL("27")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_VFMA) asm volatile(R8(VFMA("20") VFMA("21") VFMA("22") VFMA("23") VFMA("24") VFMA("25") VFMA("26") VFMA("27")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_VPKMUL) asm volatile(R8(VPKMUL("20:21") VPKMUL("22:23") VPKMUL("24:25") VPKMUL("26:27") VPKMUL("20:21") VPKMUL("22:23") VPKMUL("24:25") VPKMUL("26:27")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_VPKFMA) asm volatile(R8(VPKFMA("20:21") VPKFMA("22:23") VPKFMA("24:25") VPKFMA("26:27") VPKFMA("20:21") VPKFMA("22:23") VPKFMA("24:25") VPKFMA("26:27")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_VCVT) asm volatile(R8(VCVT("20") VCVT("21") VCVT("22") VCVT("23") VCVT("24") VCVT("25") VCVT("26") VCVT("27")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_VPERM) asm volatile(R8(VPERM("20") VPERM("21") VPERM("22") VPERM("23") VPERM("24") VPERM("25") VPERM("26") VPERM("27")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_VANDOR) asm volatile(R8(VANDOR("20") VANDOR("21") VANDOR("22") VANDOR("23") VANDOR("24") VANDOR("25") VANDOR("26") VANDOR("27")) ::"v"(lds_addr) : CLOB_V);
    // 16 reads in flight, then a wait: 64 per body
    if constexpr (B == B_DSR64) asm volatile(R4(DSR64("30:31","0") DSR64("32:33","512") DSR64("34:35","1024") DSR64("36:37","1536") DSR64("38:39","2048") DSR64("40:41","2560") DSR64("42:43","3072") DSR64("44:45","3584")
        DSR64("46:47","4096") DSR64("48:49","4608") DSR64("50:51","5120") DSR64("52:53","5632") DSR64("54:55","6144") DSR64("56:57","6656") DSR64("58:59","7168") DSR64("60:61","7680") WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_DSR32) asm volatile(R4(DSR32("30","0") DSR32("32","512") DSR32("34","1024") DSR32("36","1536") DSR32("38","2048") DSR32("40","2560") DSR32("42","3072") DSR32("44","3584")
        DSR32("46","4096") DSR32("48","4608") DSR32("50","5120") DSR32("52","5632") DSR32("54","6144") DSR32("56","6656") DSR32("58","7168") DSR32("60","7680") WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_DSR128) asm volatile(R8(DSR128("30:33","0") DSR128("34:37","1024") DSR128("38:41","2048") DSR128("42:45","3072") DSR128("46:49","4096") DSR128("50:53","5120") DSR128("54:57","6144") DSR128("58:61","7168") WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_DSW128) asm volatile(R8(DSW128("0") DSW128("1024") DSW128("2048") DSW128("3072") DSW128("4096") DSW128("5120") DSW128("6144") DSW128("7168") WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_DSW64) asm volatile(R8(DSW64("0") DSW64("512") DSW64("1024") DSW64("1536") DSW64("2048") DSW64("2560") DSW64("3072") DSW64("3584") WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_1) asm volatile(R8(M32("64:79") M32("64:79") M32("64:79") M32("64:79")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_2) asm volatile(R8(M32("64:79") M32("80:95") M32("64:79") M32("80:95")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_4) asm volatile(R8(M32("64:79") M32("80:95") M32("96:111") M32("112:127")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M16_1) asm volatile(R8(M16("64:67") M16("64:67") M16("64:67") M16("64:67") M16("64:67") M16("64:67") M16("64:67") M16("64:67")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M16_2) asm volatile(R8(M16("64:67") M16("68:71") M16("64:67") M16("68:71") M16("64:67") M16("68:71") M16("64:67") M16("68:71")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M16_4) asm volatile(R8(M16("64:67") M16("68:71") M16("72:75") M16("76:79") M16("64:67") M16("68:71") M16("72:75") M16("76:79")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M16_8) asm volatile(R8(M16("64:67") M16("68:71") M16("72:75") M16("76:79") M16("80:83") M16("84:87") M16("88:91") M16("92:95")) ::"v"(lds_addr) : CLOB_V);
    // one 32x32x16 MFMA (4 accumulators in rotation) + k VALU
#define V2 VMUL("20") VPERM("21")
#define V4 V2 VCVT("22") VMUL("23")
#define V6 V4 VPERM("24") VCVT("25")
#define V8 V6 VMUL("26") VMUL("27")
#define MIX4(v) M32("64:79") v M32("80:95") v M32("96:111") v M32("112:127") v
    if constexpr (B == B_M32_4_V2) asm volatile(R8(MIX4(V2)) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_4_V4) asm volatile(R8(MIX4(V4)) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_4_V6) asm volatile(R8(MIX4(V6)) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_4_V8) asm volatile(R8(MIX4(V8)) ::"v"(lds_addr) : CLOB_V);
    // ... + 1 / 2 LDS reads per MFMA (results never waited for inside the group of four; one wait per group)
#define L1a DSR64("30:31","0")
#define L1b DSR64("32:33","512")
#define L1c DSR64("34:35","1024")
#define L1d DSR64("36:37","1536")
    if constexpr (B == B_M32_4_V6_L1) asm volatile(R8(M32("64:79") V6 L1a M32("80:95") V6 L1b M32("96:111") V6 L1c M32("112:127") V6 L1d WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_4_V6_L2) asm volatile(R8(M32("64:79") V6 L1a DSR128("38:41","2048") M32("80:95") V6 L1b DSR128("42:45","3072") M32("96:111") V6 L1c DSR128("46:49","4096") M32("112:127") V6 L1d DSR128("50:53","5120") WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_2_V6_L2) asm volatile(R8(M32("64:79") V6 L1a DSR128("38:41","2048") M32("80:95") V6 L1b DSR128("42:45","3072") M32("64:79") V6 L1c DSR128("46:49","4096") M32("80:95") V6 L1d DSR128("50:53","5120") WAITL) ::"v"(lds_addr) : CLOB_V);
#define MIX2(v) M32("64:79") v M32("80:95") v M32("64:79") v M32("80:95") v
    if constexpr (B == B_M32_2_V6) asm volatile(R8(MIX2(V6)) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_2_V2) asm volatile(R8(MIX2(V2)) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_1_V2) asm volatile(R8(M32("64:79") V2 M32("64:79") V2 M32("64:79") V2 M32("64:79") V2) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_2_L1) asm volatile(R8(M32("64:79") L1a M32("80:95") L1b M32("64:79") L1c M32("80:95") L1d WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_2_L2) asm volatile(R8(M32("64:79") L1a DSR128("38:41","2048") M32("80:95") L1b DSR128("42:45","3072") M32("64:79") L1c DSR128("46:49","4096") M32("80:95") L1d DSR128("50:53","5120") WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M32_4_L2) asm volatile(R8(M32("64:79") L1a DSR128("38:41","2048") M32("80:95") L1b DSR128("42:45","3072") M32("96:111") L1c DSR128("46:49","4096") M32("112:127") L1d DSR128("50:53","5120") WAITL) ::"v"(lds_addr) : CLOB_V);
    // two accumulators, the two MFMAs of a step back to back, then the VALU / LDS work of the step
    if constexpr (B == B_M32_2x2_V6_L2) asm volatile(R8(M32("64:79") M32("80:95") V6 V6 L1a DSR128("38:41","2048") L1b DSR128("42:45","3072") M32("64:79") M32("80:95") V6 V6 L1c DSR128("46:49","4096") L1d DSR128("50:53","5120") WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M16_2_V2) asm volatile(R8(M16("64:67") V2 M16("68:71") V2 M16("64:67") V2 M16("68:71") V2) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M16_4_V3_L1) asm volatile(R8(M16("64:67") V2 VFMA("22") L1a M16("68:71") V2 VFMA("23") L1b M16("72:75") V2 VFMA("24") L1c M16("76:79") V2 VFMA("25") L1d WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_VMUL16) asm volatile(R4(VMUL("20") VMUL("21") VMUL("22") VMUL("23") VMUL("24") VMUL("25") VMUL("26") VMUL("27") VMUL("30") VMUL("31") VMUL("32") VMUL("33") VMUL("34") VMUL("35") VMUL("36") VMUL("37")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_VMULSAME) asm volatile(R8("v_mul_f32 v20, v10, v10\n v_mul_f32 v21, v10, v10\n v_mul_f32 v22, v10, v10\n v_mul_f32 v23, v10, v10\n v_mul_f32 v24, v10, v10\n v_mul_f32 v25, v10, v10\n v_mul_f32 v26, v10, v10\n v_mul_f32 v27, v10, v10\n") ::"v"(lds_addr) : CLOB_V);
    // write-after-read against the sources of an MFMA just issued: 6 VALU per MFMA writing (0) unrelated registers, (1) the B
    // operand of that MFMA, (2) the B operand of the MFMA issued before it, (3) ... two before it; (4) an LDS read returning
    // into the A operand of that MFMA. Operand sets rotate over four register groups: A v[10:13] fixed, B v[14:17] / v[28:31] / v[54:57] / v[58:61]
#define MB(acc, b) "v_mfma_f32_32x32x16_bf16 v[" acc "], v[10:13], v[" b "], v[" acc "]\n"
#define W4(a, b, c, d) "v_mul_f32 v" a ", v18, v19\n v_perm_b32 v" b ", v18, v19, v20\n v_cvt_pk_bf16_f32 v" c ", v18, v19\n v_mul_f32 v" d ", v18, v19\n v_mul_f32 v21, v18, v19\n v_mul_f32 v22, v18, v19\n"
    if constexpr (B == B_WAR0) asm volatile(R8(MB("64:79","14:17") W4("23","24","25","26") MB("80:95","28:31") W4("23","24","25","26") MB("96:111","54:57") W4("23","24","25","26") MB("112:127","58:61") W4("23","24","25","26")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_WAR1) asm volatile(R8(MB("64:79","14:17") W4("14","15","16","17") MB("80:95","28:31") W4("28","29","30","31") MB("96:111","54:57") W4("54","55","56","57") MB("112:127","58:61") W4("58","59","60","61")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_WAR2) asm volatile(R8(MB("64:79","14:17") W4("58","59","60","61") MB("80:95","28:31") W4("14","15","16","17") MB("96:111","54:57") W4("28","29","30","31") MB("112:127","58:61") W4("54","55","56","57")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_WAR3) asm volatile(R8(MB("64:79","14:17") W4("54","55","56","57") MB("80:95","28:31") W4("58","59","60","61") MB("96:111","54:57") W4("14","15","16","17") MB("112:127","58:61") W4("28","29","30","31")) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_WAR4) asm volatile(R8(MB("64:79","14:17") "ds_read_b128 v[14:17], %0\n" W4("23","24","25","26") MB("80:95","28:31") "ds_read_b128 v[28:31], %0 offset:1024\n" W4("23","24","25","26") MB("96:111","54:57") "ds_read_b128 v[54:57], %0 offset:2048\n" W4("23","24","25","26") MB("112:127","58:61") "ds_read_b128 v[58:61], %0 offset:3072\n" W4("23","24","25","26")) ::"v"(lds_addr) : CLOB_V);
#define MIX16_4(v) M16("64:67") v M16("68:71") v M16("72:75") v M16("76:79") v
    if constexpr (B == B_M16_4_V2) asm volatile(R8(MIX16_4(V2)) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M16_4_V4) asm volatile(R8(MIX16_4(V4)) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_M16_8_V3_L1) asm volatile(R4(M16("64:67") V2 VFMA("22") L1a M16("68:71") V2 VFMA("23") L1b M16("72:75") V2 VFMA("24") L1c M16("76:79") V2 VFMA("25") L1d
                                                    M16("80:83") V2 VFMA("22") L1a M16("84:87") V2 VFMA("23") L1b M16("88:91") V2 VFMA("24") L1c M16("92:95") V2 VFMA("25") L1d WAITL) ::"v"(lds_addr) : CLOB_V);
    // decode option A per byte: perm, ds_read_b64, pk_mul, cvt_pk (independent streams: 8 bytes per group)
#define DA(i, off) VPERM("2" i) DSR64("3" i ":3" i, off)
    if constexpr (B == B_DECODE_A) asm volatile(R8(VPERM("20") DSR64("30:31","0") VPERM("21") DSR64("32:33","512") VPERM("22") DSR64("34:35","1024") VPERM("23") DSR64("36:37","1536")
        VPERM("24") DSR64("38:39","2048") VPERM("25") DSR64("40:41","2560") VPERM("26") DSR64("42:43","3072") VPERM("27") DSR64("44:45","3584")
        VPKMUL("46:47") VCVT("20") VPKMUL("48:49") VCVT("21") VPKMUL("50:51") VCVT("22") VPKMUL("52:53") VCVT("23") VPKMUL("54:55") VCVT("24") VPKMUL("56:57") VCVT("25") VPKMUL("58:59") VCVT("26") VPKMUL("60:61") VCVT("27") WAITL) ::"v"(lds_addr) : CLOB_V);
    if constexpr (B == B_DECODE_D) asm volatile(R8(VPERM("20") DSR32("30","0") VPERM("21") DSR32("32","512") VPERM("22") DSR32("34","1024") VPERM("23") DSR32("36","1536")
        VPERM("24") DSR32("38","2048") VPERM("25") DSR32("40","2560") VPERM("26") DSR32("42","3072") VPERM("27") DSR32("44","3584") WAITL) ::"v"(lds_addr) : CLOB_V);
}

static const Body kBodies[B_COUNT] = {
    {"idle", 1}, {"v_mul_f32", 64}, {"v_fma_f32", 64}, {"v_pk_mul_f32", 64}, {"v_pk_fma_f32", 64}, {"v_cvt_pk_bf16_f32", 64}, {"v_perm_b32", 64}, {"v_and_or_b32", 64},
    {"ds_read_b64 (16 in flight)", 64}, {"ds_read_b32 (16 in flight)", 64}, {"ds_read_b128 (8 in flight)", 64}, {"ds_write_b128", 64}, {"ds_write_b64", 64},
    {"mfma32 1 acc", 32}, {"mfma32 2 acc", 32}, {"mfma32 4 acc", 32}, {"mfma16 1 acc", 64}, {"mfma16 2 acc", 64}, {"mfma16 4 acc", 64}, {"mfma16 8 acc", 64},
    {"mfma32(4) + 2 valu", 32}, {"mfma32(4) + 4 valu", 32}, {"mfma32(4) + 6 valu", 32}, {"mfma32(4) + 8 valu", 32}, {"mfma32(4) + 6 valu + b64", 32}, {"mfma32(4) + 6 valu + b64 + b128", 32},
    {"mfma32(2) + 6 valu + b64 + b128", 32}, {"mfma32(2) + 6 valu", 32}, {"mfma32(2) + 2 valu", 32}, {"mfma32(1) + 2 valu", 32}, {"mfma32(2) + b64", 32}, {"mfma32(2) + b64 + b128", 32}, {"mfma32(4) + b64 + b128", 32}, {"mfma32 pairs(2) + 12 valu + 2 b64 + 2 b128", 32}, {"mfma16(2) + 2 valu", 32}, {"mfma16(4) + 3 valu + b64", 32}, {"v_mul_f32 16 dsts", 64}, {"v_mul_f32 same src", 64}, {"mfma32 + 6 valu writing unrelated regs", 32}, {"mfma32 + 6 valu writing ITS B operand", 32}, {"mfma32 + 6 valu writing the B of the MFMA before", 32}, {"mfma32 + 6 valu writing the B of two before", 32}, {"mfma32 + ds_read_b128 into ITS B operand + 6 valu", 32},
    {"mfma16(4) + 2 valu", 32}, {"mfma16(4) + 4 valu", 32}, {"mfma16(8) + 3 valu + b64", 32}, {"decode A: perm,b64,pkmul,cvt per byte", 64}, {"decode D: perm,b32 per byte", 64},
};

// waves 0-3 and (when WPS >= 2) 4-7, 8-11, 12-15 run body BA; REPS repetitions inside the bracket
template <int BA, int BB>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int reps, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // bank-private addressing inside a 512-byte row: lane * 8 (b64), * 4 (b32) both conflict-free; b128 lane * 16 spans 1 KiB
    uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem)) + wave * 8192;
    if (BA == B_DSR128 || BA == B_DSW128 || BA == B_M32_4_V6_L2 || BA == B_WAR4)
        addr += lane * 16;
    else if (BA == B_DSR32 || BA == B_DECODE_D)
        addr += lane * 4;
    else
        addr += lane * 8;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((wave >> 2) % 2 == 0 || BB < 0) {
        for (int r = 0; r < reps; ++r)
            body<BA>(addr);
    } else {
        for (int r = 0; r < reps; ++r)
            body < BB < 0 ? 0 : BB > (addr);
    }
    asm volatile("s_nop 7\ns_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0)
        { out[wave] = t0; out[16 + wave] = t1; }
}

template <int BA, int BB = -1> void run(int waves_per_simd, unsigned long long* d) {
    const int reps = 64, nw = 4 * waves_per_simd;
    auto kern = k<BA, BB>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    std::vector<unsigned long long> h(32);
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * nw), 131072, 0, d, reps, nw);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, 32 * 8, hipMemcpyDeviceToHost);
    // span of a group = last end - first start over its wavefronts (arbitration is oldest-first, not round-robin)
    unsigned long long lo0 = ~0ull, lo1 = 0, hi0 = ~0ull, hi1 = 0;
    for (int w = 0; w < nw; ++w) {
        if (BB >= 0 && (w >> 2) % 2 == 1) { hi0 = std::min(hi0, h[w]); hi1 = std::max(hi1, h[16 + w]); } else { lo0 = std::min(lo0, h[w]); lo1 = std::max(lo1, h[16 + w]); }
    }
    const double lo = double(lo1 - lo0) / reps;
    if (BB < 0) {
        const double per = lo / kBodies[BA].count;
        printf("%-44s %d wave/SIMD: span %7.1f cycles per instruction = %6.2f per SIMD and instruction\n", kBodies[BA].name, waves_per_simd, per, per / waves_per_simd);
    } else {
        const double hi = double(hi1 - hi0) / reps;
        printf("%-44s | %-36s %d wave/SIMD: %7.1f | %7.1f cycles per body (span)\n", kBodies[BA].name, kBodies[BB].name, waves_per_simd, lo, hi);
    }
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 32 * 8);
#define ALLW(B) run<B>(1, d); run<B>(2, d); run<B>(4, d);
    ALLW(B_VMUL) ALLW(B_VFMA) ALLW(B_VPKMUL) ALLW(B_VPKFMA) ALLW(B_VCVT) ALLW(B_VPERM) ALLW(B_VANDOR)
    ALLW(B_DSR64) ALLW(B_DSR32) ALLW(B_DSR128) ALLW(B_DSW128) ALLW(B_DSW64)
    ALLW(B_M32_1) ALLW(B_M32_2) ALLW(B_M32_4) ALLW(B_M16_1) ALLW(B_M16_2) ALLW(B_M16_4) ALLW(B_M16_8)
    ALLW(B_M32_4_V2) ALLW(B_M32_4_V4) ALLW(B_M32_4_V6) ALLW(B_M32_4_V8) ALLW(B_M32_4_V6_L1) ALLW(B_M32_4_V6_L2) ALLW(B_M32_2_V6_L2) ALLW(B_M32_2_V6) ALLW(B_M32_2_V2) ALLW(B_M32_1_V2) ALLW(B_M32_2_L1) ALLW(B_M32_2_L2) ALLW(B_M32_4_L2) ALLW(B_M32_2x2_V6_L2) ALLW(B_M16_2_V2) ALLW(B_M16_4_V3_L1) ALLW(B_VMUL16) ALLW(B_VMULSAME) ALLW(B_WAR0) ALLW(B_WAR1) ALLW(B_WAR2) ALLW(B_WAR3) ALLW(B_WAR4)
    ALLW(B_M16_4_V2) ALLW(B_M16_4_V4) ALLW(B_M16_8_V3_L1) ALLW(B_DECODE_A) ALLW(B_DECODE_D)
    // pairs: first group of four wavefronts | second group (same SIMDs)
    run<B_M32_4, B_VMUL>(2, d); run<B_M32_4, B_DECODE_A>(2, d); run<B_M32_2, B_DECODE_A>(2, d); run<B_M32_4, B_DSR64>(2, d);
    run<B_M32_4_V6_L2, B_M32_4_V6_L2>(2, d);
    return 0;
}
