// rtk_simulate: seeded synthetic data generator for the BASELINE.json configs.
//
// Writes  PREFIX.ref.fa   reference (one record per haplotype when --het > 0)
//         PREFIX.sr.fq    paired-end short reads, interleaved, both mates carry the same name
//                         (reference README: "reads from the same pair must have the same name")
//         PREFIX.lr.fq    long reads, 4-line FASTQ
// There is no counterpart in the reference (it ships no data and no generator); everything is
// derived from one 64-bit seed so every box regenerates identical inputs.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../common/kmer.hpp"

using namespace rtk;

struct Opt {
    std::string prefix = "sim";
    uint64_t seed = 1;
    size_t ref_len = 50000;
    double het = 0.0;          // heterozygous SNP rate between the two haplotypes (0 = haploid)
    size_t sr_pairs = 0;       // if 0 use sr_cov
    double sr_cov = 30.0;
    size_t sr_len = 150;
    double sr_err = 0.001;
    double ins_mean = 500.0, ins_sd = 50.0;
    size_t lr_n = 0;           // if 0 use lr_cov
    double lr_cov = 30.0;
    size_t lr_len = 10000;     // fixed length (profile uniform) or median (profile ont)
    double lr_err = 0.10;
    bool lr_truth = false;
    std::string lr_profile = "uniform"; // uniform: sub:ins:del = 4:3:3 ; ont: 35:25:40, log-normal length, homopolymer-biased indels
    double repeat_frac = 0.0;  // fraction of the reference made of two-copy repeats (config 5)
    size_t tandem = 0;         // number of tandem-repeat blocks (unit 7..45 bp, spanning > 2k): short cycles in the graph
};

static double gauss(Rng& r) {
    double u1 = r.uniform(), u2 = r.uniform();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
}

static std::string mutate_long(const std::string& src, const Opt& o, Rng& rng) {
    const bool ont = (o.lr_profile == "ont");
    const double p_sub = ont ? 0.35 : 0.4, p_ins = ont ? 0.25 : 0.3; // remainder = deletion
    std::string out;
    out.reserve(src.size() + src.size() / 8);
    for (size_t i = 0; i < src.size(); ++i) {
        double perr = o.lr_err;
        bool homop = false;
        if (ont && i >= 2 && src[i] == src[i - 1] && src[i] == src[i - 2]) { homop = true; perr *= 1.5; }
        if (rng.uniform() < perr) {
            double t = rng.uniform();
            double ps = p_sub, pi = p_ins;
            if (homop) { ps = 0.15; pi = 0.35; } // homopolymer runs: mostly indels
            if (t < ps) { char c; do { c = bits2base(static_cast<int>(rng.below(4))); } while (c == src[i]); out.push_back(c); }
            else if (t < ps + pi) { out.push_back(src[i]); out.push_back(homop ? src[i] : bits2base(static_cast<int>(rng.below(4)))); }
            else { /* deletion */ }
        } else out.push_back(src[i]);
    }
    return out;
}

int main(int argc, char** argv) {
    Opt o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char* n) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "rtk_simulate: missing value for %s\n", n); exit(2); } return argv[++i]; };
        if (a == "--prefix") o.prefix = need("--prefix");
        else if (a == "--seed") o.seed = strtoull(need("--seed"), nullptr, 10);
        else if (a == "--ref-len") o.ref_len = strtoull(need("--ref-len"), nullptr, 10);
        else if (a == "--het") o.het = atof(need("--het"));
        else if (a == "--sr-pairs") o.sr_pairs = strtoull(need("--sr-pairs"), nullptr, 10);
        else if (a == "--sr-cov") o.sr_cov = atof(need("--sr-cov"));
        else if (a == "--sr-len") o.sr_len = strtoull(need("--sr-len"), nullptr, 10);
        else if (a == "--sr-err") o.sr_err = atof(need("--sr-err"));
        else if (a == "--insert-mean") o.ins_mean = atof(need("--insert-mean"));
        else if (a == "--insert-sd") o.ins_sd = atof(need("--insert-sd"));
        else if (a == "--lr-n") o.lr_n = strtoull(need("--lr-n"), nullptr, 10);
        else if (a == "--lr-cov") o.lr_cov = atof(need("--lr-cov"));
        else if (a == "--lr-len") o.lr_len = strtoull(need("--lr-len"), nullptr, 10);
        else if (a == "--lr-err") o.lr_err = atof(need("--lr-err"));
        else if (a == "--lr-profile") o.lr_profile = need("--lr-profile");
        else if (a == "--lr-truth") o.lr_truth = true;
        else if (a == "--repeat-frac") o.repeat_frac = atof(need("--repeat-frac"));
        else if (a == "--tandem") o.tandem = strtoull(need("--tandem"), nullptr, 10);
        else { fprintf(stderr, "rtk_simulate: unknown option %s\n", a.c_str()); return 2; }
    }
    Rng rng(o.seed);

    // --- reference (haplotype 0), optional two-copy repeats, optional second haplotype ---
    std::string hap0(o.ref_len, 'A');
    for (size_t i = 0; i < o.ref_len; ++i) hap0[i] = bits2base(static_cast<int>(rng.below(4)));
    if (o.repeat_frac > 0.0) {
        const size_t rep_len = 2000;
        const size_t n_rep = static_cast<size_t>(o.repeat_frac * o.ref_len / (2.0 * rep_len));
        for (size_t r = 0; r < n_rep && o.ref_len > 4 * rep_len; ++r) {
            const size_t a = rng.below(o.ref_len - rep_len), b = rng.below(o.ref_len - rep_len);
            if (a + rep_len <= b || b + rep_len <= a) hap0.replace(b, rep_len, hap0, a, rep_len);
        }
    }
    std::vector<std::pair<size_t, size_t> > tandems; // (position, unit length)
    for (size_t t = 0; t < o.tandem && o.ref_len > 2000; ++t) {
        const size_t unit = 7 + rng.below(39), copies = 3 + (62 + unit - 1) / unit, span = unit * copies;
        const size_t pos = 500 + rng.below(o.ref_len - span - 1000);
        for (size_t i = unit; i < span; ++i) hap0[pos + i] = hap0[pos + i - unit];
        tandems.push_back(std::make_pair(pos, unit));
    }
    std::vector<std::string> haps(1, hap0);
    if (o.het > 0.0) {
        std::string h1 = hap0;
        for (size_t i = 0; i < h1.size(); ++i) if (rng.uniform() < o.het) { char c; do { c = bits2base(static_cast<int>(rng.below(4))); } while (c == h1[i]); h1[i] = c; }
        // the second haplotype loses one unit of every other tandem block: reads and graph paths then disagree in copy number
        std::vector<std::pair<size_t, size_t> > td = tandems; std::sort(td.begin(), td.end());
        for (size_t t = td.size(); t-- > 0;) if (t % 2 == 0) h1.erase(td[t].first, td[t].second);
        haps.push_back(h1);
    }
    {
        FILE* f = fopen((o.prefix + ".ref.fa").c_str(), "w");
        if (!f) { perror("rtk_simulate: ref"); return 1; }
        for (size_t h = 0; h < haps.size(); ++h) { fprintf(f, ">hap%zu\n", h); fwrite(haps[h].data(), 1, haps[h].size(), f); fputc('\n', f); } // (not "%s": printf counts in int, a 3 Gb haplotype came out as 2^32 bytes)
        fclose(f);
    }

    // --- short reads ---
    {
        const size_t n_pairs = o.sr_pairs ? o.sr_pairs : static_cast<size_t>(o.sr_cov * o.ref_len / (2.0 * o.sr_len));
        FILE* f = fopen((o.prefix + ".sr.fq").c_str(), "w");
        if (!f) { perror("rtk_simulate: sr"); return 1; }
        const std::string q(o.sr_len, 'I');
        for (size_t p = 0; p < n_pairs; ++p) {
            const size_t hap_i = rng.below(haps.size());
            const std::string& hap = haps[hap_i];
            size_t ins = static_cast<size_t>(std::max(static_cast<double>(o.sr_len), o.ins_mean + o.ins_sd * gauss(rng)));
            if (ins > hap.size()) ins = hap.size();
            const size_t start = rng.below(hap.size() - ins + 1);
            std::string frag = hap.substr(start, ins);
            if (rng.below(2)) frag = reverse_complement(frag);
            std::string m1 = frag.substr(0, o.sr_len);
            std::string m2 = reverse_complement(frag).substr(0, o.sr_len);
            for (int m = 0; m < 2; ++m) {
                std::string& s = m ? m2 : m1;
                for (size_t i = 0; i < s.size(); ++i) if (rng.uniform() < o.sr_err) { char c; do { c = bits2base(static_cast<int>(rng.below(4))); } while (c == s[i]); s[i] = c; }
                fprintf(f, "@sr%zu\n%s\n+\n%s\n", p, s.c_str(), q.substr(0, s.size()).c_str());
            }
        }
        fclose(f);
    }

    // --- long reads ---
    {
        FILE* f = fopen((o.prefix + ".lr.fq").c_str(), "w");
        if (!f) { perror("rtk_simulate: lr"); return 1; }
        const bool ont = (o.lr_profile == "ont");
        FILE* ft = o.lr_truth ? fopen((o.prefix + ".lr.truth.tsv").c_str(), "w") : nullptr; // --lr-truth: for size-independent checks of big runs (profiles/scripts/config4_dry_run.py)
        size_t total = 0, n = 0;
        const size_t target_bases = static_cast<size_t>(o.lr_cov * o.ref_len);
        while (o.lr_n ? (n < o.lr_n) : (total < target_bases)) {
            const size_t hap_i = rng.below(haps.size());
            const std::string& hap = haps[hap_i];
            size_t len = o.lr_len;
            if (ont) {
                const double l = std::exp(std::log(static_cast<double>(o.lr_len)) + 0.6 * gauss(rng));
                len = static_cast<size_t>(std::min(100000.0, std::max(1000.0, l)));
            }
            if (len > hap.size()) len = hap.size();
            const size_t start = rng.below(hap.size() - len + 1);
            std::string s = hap.substr(start, len);
            const bool rev = rng.below(2) != 0;
            if (rev) s = reverse_complement(s);
            if (ft) fprintf(ft, "lr%zu\t%zu\t%zu\t%zu\t%c\n", n, hap_i, start, len, rev ? '-' : '+'); // where the read comes from (haplotype record of ref.fa, 0-based start, length, strand)
            s = mutate_long(s, o, rng);
            std::string q(s.size(), '5');
            for (size_t i = 0; i < q.size(); ++i) q[i] = static_cast<char>(33 + 5 + rng.below(20));
            fprintf(f, "@lr%zu\n%s\n+\n%s\n", n, s.c_str(), q.c_str());
            total += len; ++n;
        }
        fclose(f); if (ft) fclose(ft);
        fprintf(stderr, "rtk_simulate: %zu long reads, %zu source bases\n", n, total);
    }
    return 0;
}
