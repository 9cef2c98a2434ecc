/*!
 * \file hip_mat5_writer.h
 * \brief The .mat files gnss-sdr's blocks leave behind (tracking: dll_pll_veml_tracking::save_matfile, trk.cc:1706-1890; acquisition:
 *        pcps_acquisition::dump_results, acq.cc:354-406), written without matio.
 *
 * The reference writes them through matio as MAT 7.3 (an HDF5 container).  This is a plain MAT-file Level 5 writer -- the published
 * format ("MAT-File Format", MathWorks): a 128-byte header, then one miMATRIX element per variable (array flags, dimensions, name,
 * real part), little endian, uncompressed, every sub-element padded to 8 bytes.  MATLAB's load(), Octave and scipy.io.loadmat read it;
 * the variables (names, classes, dimensions, values) are the reference's.  Column-major data, as MATLAB stores it.
 */
#ifndef GNSS_SDR_HIP_MAT5_WRITER_H
#define GNSS_SDR_HIP_MAT5_WRITER_H

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

class Hip_Mat5_Writer
{
public:
    explicit Hip_Mat5_Writer(const std::string& path) : d_f(std::fopen(path.c_str(), "wb"))
    {
        if (d_f == nullptr) return;
        char header[128];
        std::memset(header, ' ', sizeof(header));
        static const char text[] = "MATLAB 5.0 MAT-file, Platform: gnss-sdr MI355X engine";
        std::memcpy(header, text, sizeof(text) - 1);
        std::memset(header + 116, 0, 8);  // subsystem data offset: none
        const uint16_t version = 0x0100, endian = 0x4D49;  // 'M' 'I' read as a little-endian 16-bit word: written "IM"
        std::memcpy(header + 124, &version, 2);
        std::memcpy(header + 126, &endian, 2);
        put(header, sizeof(header));
    }
    ~Hip_Mat5_Writer() { close(); }
    Hip_Mat5_Writer(const Hip_Mat5_Writer&) = delete;
    Hip_Mat5_Writer& operator=(const Hip_Mat5_Writer&) = delete;
    bool ok() const { return d_f != nullptr && !d_failed; }
    bool close()
    {
        if (d_f != nullptr)
            {
                if (std::fclose(d_f) != 0) d_failed = true;
                d_f = nullptr;
            }
        return !d_failed;
    }

    /*! a rows x cols array of T (float, double, int32_t, uint32_t, uint64_t), column-major */
    template <typename T>
    void matrix(const std::string& name, const T* data, size_t rows, size_t cols)
    {
        const uint32_t n_bytes = static_cast<uint32_t>(rows * cols * sizeof(T));
        const uint32_t name_len = static_cast<uint32_t>(name.size());
        const uint32_t body = 16 + 16 + (8 + pad8(name_len)) + (8 + pad8(n_bytes));
        tag(MI_MATRIX, body);
        // array flags: class in the low byte of the first word, no complex / global / logical flags
        tag(MI_UINT32, 8);
        const uint32_t flags[2] = {class_of<T>(), 0U};
        put(flags, 8);
        tag(MI_INT32, 8);
        const int32_t dims[2] = {static_cast<int32_t>(rows), static_cast<int32_t>(cols)};
        put(dims, 8);
        tag(MI_INT8, name_len);
        put(name.data(), name_len);
        pad(name_len);
        tag(type_of<T>(), n_bytes);
        put(data, n_bytes);
        pad(n_bytes);
    }
    template <typename T>
    void scalar(const std::string& name, T value)
    {
        matrix(name, &value, 1, 1);
    }

private:
    enum : uint32_t
    {
        MI_INT8 = 1,
        MI_INT32 = 5,
        MI_UINT32 = 6,
        MI_SINGLE = 7,
        MI_DOUBLE = 9,
        MI_UINT64 = 13,
        MI_MATRIX = 14
    };
    template <typename T>
    static uint32_t class_of()
    {
        if (std::is_same<T, double>::value) return 6;    // mxDOUBLE_CLASS
        if (std::is_same<T, float>::value) return 7;     // mxSINGLE_CLASS
        if (std::is_same<T, int32_t>::value) return 12;  // mxINT32_CLASS
        if (std::is_same<T, uint32_t>::value) return 13; // mxUINT32_CLASS
        static_assert(std::is_same<T, double>::value || std::is_same<T, float>::value || std::is_same<T, int32_t>::value || std::is_same<T, uint32_t>::value ||
                          std::is_same<T, uint64_t>::value,
            "unsupported element type");
        return 15;  // mxUINT64_CLASS
    }
    template <typename T>
    static uint32_t type_of()
    {
        if (std::is_same<T, double>::value) return MI_DOUBLE;
        if (std::is_same<T, float>::value) return MI_SINGLE;
        if (std::is_same<T, int32_t>::value) return MI_INT32;
        if (std::is_same<T, uint32_t>::value) return MI_UINT32;
        return MI_UINT64;
    }
    static uint32_t pad8(uint32_t n) { return (n + 7U) & ~7U; }
    void tag(uint32_t type, uint32_t bytes)
    {
        const uint32_t t[2] = {type, bytes};
        put(t, 8);
    }
    void pad(uint32_t n)
    {
        static const char zeros[8] = {0};
        put(zeros, pad8(n) - n);
    }
    void put(const void* p, size_t n)
    {
        if (d_f == nullptr || n == 0) return;
        if (std::fwrite(p, 1, n, d_f) != n) d_failed = true;
    }
    std::FILE* d_f;
    bool d_failed{false};
};

/*! dll_pll_veml_tracking::save_matfile (trk.cc:1706-1890): the block's binary dump (108-byte records: five magnitudes, Prompt_I/Q, sample counter,
    twelve loop values, two auxiliaries, PRN, TOW, week) as one 1 x N variable per field in <dump file without ".dat">.mat.  Returns the number of
    epochs written, -1 when a file cannot be read / written. */
inline long hip_tracking_dump_to_mat(const std::string& dat_path)
{
    std::FILE* f = std::fopen(dat_path.c_str(), "rb");
    if (f == nullptr) return -1;
    constexpr size_t REC = sizeof(uint64_t) + sizeof(double) + 19 * sizeof(float) + sizeof(uint32_t) + sizeof(uint64_t) + sizeof(int32_t);  // trk.cc:1708-1713
    static_assert(REC == 108, "log_data record");
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const size_t n = bytes > 0 ? static_cast<size_t>(bytes) / REC : 0;
    // field order of log_data (trk.cc:1599-1702) = read order of save_matfile (:1772-1795)
    static const char* const float_names[19] = {"abs_VE", "abs_E", "abs_P", "abs_L", "abs_VL", "Prompt_I", "Prompt_Q", "acc_carrier_phase_rad", "carrier_doppler_hz",
        "carrier_doppler_rate_hz", "code_freq_chips", "code_freq_rate_chips", "carr_error_hz", "carr_error_filt_hz", "code_error_chips", "code_error_filt_chips",
        "CN0_SNV_dB_Hz", "carrier_lock_test", "aux1"};
    std::vector<std::vector<float>> fl(19, std::vector<float>(n));
    std::vector<uint64_t> start(n), tow(n);
    std::vector<double> aux2(n);
    std::vector<uint32_t> prn(n);
    std::vector<int32_t> wn(n);
    bool ok = true;
    for (size_t i = 0; i < n && ok; i++)
        {
            unsigned char rec[REC];
            ok = std::fread(rec, 1, REC, f) == REC;
            const unsigned char* p = rec;
            for (int k = 0; k < 7; k++, p += 4) std::memcpy(&fl[static_cast<size_t>(k)][i], p, 4);
            std::memcpy(&start[i], p, 8);
            p += 8;
            for (int k = 7; k < 19; k++, p += 4) std::memcpy(&fl[static_cast<size_t>(k)][i], p, 4);
            std::memcpy(&aux2[i], p, 8);
            p += 8;
            std::memcpy(&prn[i], p, 4);
            p += 4;
            std::memcpy(&tow[i], p, 8);
            p += 8;
            std::memcpy(&wn[i], p, 4);
        }
    std::fclose(f);
    if (!ok) return -1;
    std::string mat = dat_path;
    if (mat.size() >= 4) mat.erase(mat.size() - 4, 4);  // trk.cc:1812-1813
    mat.append(".mat");
    Hip_Mat5_Writer w(mat);
    if (!w.ok()) return -1;
    for (int k = 0; k < 7; k++) w.matrix(float_names[k], fl[static_cast<size_t>(k)].data(), 1, n);
    w.matrix("PRN_start_sample_count", start.data(), 1, n);
    for (int k = 7; k < 19; k++) w.matrix(float_names[k], fl[static_cast<size_t>(k)].data(), 1, n);
    w.matrix("aux2", aux2.data(), 1, n);
    w.matrix("PRN", prn.data(), 1, n);
    w.matrix("TOW_ms", tow.data(), 1, n);
    w.matrix("WN", wn.data(), 1, n);
    return w.close() ? static_cast<long>(n) : -1;
}
#endif
