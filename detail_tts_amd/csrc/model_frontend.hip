// Prompt front-end on the device: resample + log-mel spectrogram of the reference audio (api.py:34-45).
#include "model.h"

namespace dtts {

// y[b][q*neu + p] = sum_k ker[p][k] * x[b][q*orig + k - width]       (torchaudio functional.resample: conv1d, stride orig)
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int L, const float* __restrict__ ker, int orig, int neu,
                                                       int width, float* __restrict__ y, int Lout) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= Lout) return;
    const int q = i / neu, p = i - q * neu, klen = 2 * width + orig;
    const float* xb = x + (long long)b * L;
    const float* kp = ker + (long long)p * klen;
    const int j0 = q * orig - width;
    float acc = 0.f;
    for (int k = 0; k < klen; ++k) {
        const int j = j0 + k;
        if (j >= 0 && j < L) acc += kp[k] * xb[j];
    }
    y[(long long)b * Lout + i] = acc;
}

// F[b][k][t] = ypad[t*hop + k], ypad = reflect-padded row (pad = (n_fft - hop)/2 on both sides of the row's own length)
__global__ __launch_bounds__(256) void frame_reflect_kernel(const float* __restrict__ wav, int L, const int* __restrict__ lens, int n_fft,
                                                            int hop, int T, float* __restrict__ F) {
    const int t = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y, b = blockIdx.z;
    const int len = lens[b], Tb = len / hop;
    if (t >= T) return;
    float v = 0.f;
    if (t < Tb) {
        int i = t * hop + k - (n_fft - hop) / 2;
        if (i < 0) i = -i;
        if (i >= len) i = 2 * (len - 1) - i;
        v = wav[(long long)b * L + i];
    }
    F[((long long)b * n_fft + k) * T + t] = v;
}

// M[b][f][t] = sqrt(re^2 + im^2 + 1e-6), re = S[b][f][t], im = S[b][nf + f][t]     (data_utils.py:145)
__global__ __launch_bounds__(256) void magnitude_kernel(const float* __restrict__ S, int nf, int T, float* __restrict__ M) {
    const int t = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const float re = S[((long long)b * 2 * nf + f) * T + t], im = S[((long long)b * 2 * nf + nf + f) * T + t];
    M[((long long)b * nf + f) * T + t] = sqrtf(re * re + im * im + 1e-6f);
}

void Model::build_frontend() {
    // the DFT matrix is packed [n_fft + 2 rows][n_fft inputs]: recover n_fft from the packed size (CinP = n_fft, CoutP = round128(n_fft + 2))
    const size_t n = weights_.at("frontend.dft.wp").second;
    int nfft = 0;
    for (int c = 64; c <= 8192; c *= 2)
        if ((size_t)c * packed_cout(c + 2) == n) nfft = c;
    DTTS_REQUIRE(nfft > 0, "frontend.dft has an unexpected size");
    fe_nfft_ = nfft;
    fe_dft_ = conv("frontend.dft", nfft, nfft + 2, 1, false);
    fe_mel_ = conv("frontend.mel", nfft / 2 + 1, cfg.mel_channels, 1, false);
}

void Model::resample(const float* x, int B, int L, const float* kernel, int orig, int neu, int width, float* y, int Lout, hipStream_t s) {
    DTTS_REQUIRE(B > 0 && L > 0 && orig > 0 && neu > 0 && width > 0 && Lout > 0, "resample arguments");
    DTTS_REQUIRE((long long)Lout <= ((long long)neu * L + orig - 1) / orig, "resample output longer than ceil(L * new / orig)");
    hipLaunchKernelGGL(resample_kernel, dim3(cdiv(Lout, 256), B), dim3(256), 0, s, x, L, kernel, orig, neu, width, y, Lout);
    DTTS_CHECK_HIP(hipGetLastError());
}

void Model::mel_spectrogram(const float* wav, const int* lens_host, int B, int L, int n_fft, int hop, float* mel_out, int Tmax, hipStream_t s,
                            float* spec_out) {
    DTTS_REQUIRE(bound_ && has_frontend_, "front-end matrices not bound");
    DTTS_REQUIRE(n_fft == fe_nfft_ && hop > 0 && hop < n_fft && (n_fft - hop) % 2 == 0, "front-end geometry does not match the packed DFT matrix");
    const int T = L / hop, nf = n_fft / 2 + 1, pad = (n_fft - hop) / 2;
    DTTS_REQUIRE(T >= 1 && Tmax >= T, "mel output too short");
    std::vector<int> len(B), tl(B);
    for (int b = 0; b < B; ++b) {
        len[b] = lens_host ? lens_host[b] : L;
        DTTS_REQUIRE(len[b] > pad && len[b] <= L, "wav length must exceed the reflect padding");      // torch's reflect pad needs pad < len
        tl[b] = len[b] / hop;
        DTTS_REQUIRE(tl[b] >= 1, "wav shorter than one hop");
    }
    ws().ensure(sizeof(float) * ((size_t)B * n_fft * T + (size_t)B * 2 * nf * T + (size_t)B * nf * T) + 8192);
    float* F = ws().f32((size_t)B * n_fft * T);
    float* S = ws().f32((size_t)B * 2 * nf * T);
    float* M = ws().f32((size_t)B * nf * T);
    const int* dlen = upload_ints(len.data(), B, s);
    const int* dtl = upload_ints(tl.data(), B, s);
    hipLaunchKernelGGL(frame_reflect_kernel, dim3(cdiv(T, 256), n_fft, B), dim3(256), 0, s, wav, L, dlen, n_fft, hop, T, F);
    DTTS_CHECK_HIP(hipGetLastError());
    ConvParams p = cp(F, n_fft, S, 2 * nf, B, T, T, dtl);
    run_conv(fe_dft_, p, s);
    hipLaunchKernelGGL(magnitude_kernel, dim3(cdiv(T, 256), nf, B), dim3(256), 0, s, S, nf, T, M);
    DTTS_CHECK_HIP(hipGetLastError());
    if (spec_out) {          // spectrogram_torch (vqvae/utils/data_utils.py:56-87): the linear magnitudes [B, n_fft/2+1, Tmax]
        DTTS_CHECK_HIP(hipMemcpy2DAsync(spec_out, sizeof(float) * (size_t)Tmax, M, sizeof(float) * (size_t)T, sizeof(float) * (size_t)T,
                                        (size_t)B * nf, hipMemcpyDeviceToDevice, s));
        if (!mel_out) return;
    }
    ConvParams q = cp(M, nf, mel_out, cfg.mel_channels, B, T, T, dtl);
    q.y_bs = (long long)cfg.mel_channels * Tmax;
    q.y_cs = Tmax;
    q.epi_act = ACT_LOG_CLAMP;
    run_conv(fe_mel_, q, s);
}

}  // namespace dtts
