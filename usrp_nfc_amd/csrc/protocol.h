// protocol.h -- host side of the "next" row f1: closed packets -> bytes -> commands.
//
// What the reference does per packet in fsm.process_bits (fsm.py:218-238): repair the frame end
// (fsm.py:49-66), strip and check the odd parity of every ninth bit (fsm.py:28-47), find the command by the
// protocol stage of the previous command and the leading bytes (command.py:166-199, with the ISO 14443-3
// CRC_A of utilities.py:26-46 and the BCC xor check of command.py:44-67), split it into header / extra /
// CRC (command.py:245-253) and track tag type and UID (fsm.py:165-216).  Row f3, the CRYPTO1 stream cipher of
// MIFARE Classic (cipher.py, lfsr.py; fsm.py:133-154,197-213), is here too: once a Classic authentication
// starts, frames are decrypted before the parity check, nested authentications included.
// Plain host C++, no device work: a packet is a few dozen bits and the machine is sequential.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/nfc_amd.h"

namespace nfc {

// ---- ISO 14443-3 type A CRC (utilities.py:30-41): reflected 0x8408, initial 0x6363, low byte first ----
inline uint16_t crc_a(const uint8_t *data, size_t n) {
    uint32_t w = 0x6363;
    for (size_t i = 0; i < n; i++) {
        uint32_t b = data[i] ^ (w & 0xFF);
        b ^= (b << 4) & 0xFF;
        w = ((w >> 8) ^ (b << 8) ^ (b << 3) ^ (b >> 4)) & 0xFFFF;
    }
    return (uint16_t)w;
}
inline bool crc_a_ok(const uint8_t *frame, size_t n) {   // the last two bytes are the CRC
    if (n < 2) return false;
    const uint16_t c = crc_a(frame, n - 2);
    return frame[n - 2] == (c & 0xFF) && frame[n - 1] == (c >> 8);
}

// ---- the command table (command.py:78-118) ----
struct CommandDef {
    const char *name;
    int stage;
    int n_header;
    uint8_t header[2];
    int type;        // 0 tag -> reader, 1 reader -> tag (packets.py:18-20)
    int crc;         // frame ends with CRC_A
    int n_extra;
    int xor_check;   // -1: none; else the xor of the extra bytes, seeded with this, must be 0
    int total() const { return n_header + n_extra + (crc ? 2 : 0); }
};
enum {
    CMD_REQA, CMD_WUPA, CMD_ATQAUL, CMD_ATQA1K, CMD_ATQA4K, CMD_ATQADS, CMD_ANTI1R, CMD_ANTI1U, CMD_ANTI1G, CMD_SEL1R, CMD_SEL1U,
    CMD_SEL1K, CMD_ANTI2R, CMD_ANTI2T, CMD_AUTHA, CMD_AUTHB, CMD_RANDTA, CMD_RANDRB, CMD_RANDTB, CMD_SEL2R, CMD_SEL2T, CMD_READR,
    CMD_READT, CMD_HALT, CMD_WRITE, CMD_COMPW1, CMD_COMPW2, CMD_COUNT
};
static const CommandDef COMMANDS[CMD_COUNT] = {
    {"REQA", 0, 1, {0x26, 0}, 1, 0, 0, -1},
    {"WUPA", 0, 1, {0x52, 0}, 1, 0, 0, -1},
    {"ATQAUL", 0, 2, {0x44, 0x00}, 0, 0, 0, -1},
    {"ATQA1K", 0, 2, {0x04, 0x00}, 0, 0, 0, -1},
    {"ATQA1K", 0, 2, {0x02, 0x00}, 0, 0, 0, -1},   // the 4K answer carries the 1K name (command.py:84)
    {"ATQADS", 0, 2, {0x03, 0x44}, 0, 0, 0, -1},
    {"ANTI1R", 1, 2, {0x93, 0x20}, 1, 0, 0, -1},
    {"ANTI1U", 1, 1, {0x88, 0}, 0, 0, 4, 0x88},
    {"ANTI1G", 1, 0, {0, 0}, 0, 0, 5, 0},
    {"SEL1R", 2, 2, {0x93, 0x70}, 1, 1, 5, 0},
    {"SEL1U", 2, 1, {0x04, 0}, 0, 1, 0, -1},
    {"SEL1K", 2, 1, {0x08, 0}, 0, 1, 0, -1},
    {"ANTI2R", 3, 2, {0x95, 0x20}, 1, 0, 0, -1},
    {"ANTI2T", 3, 0, {0, 0}, 0, 0, 5, 0},
    {"AUTHA", 3, 1, {0x60, 0}, 1, 1, 1, -1},
    {"AUTHB", 3, 1, {0x61, 0}, 1, 1, 1, -1},
    {"RANDTA", 3, 0, {0, 0}, 0, 0, 4, -1},
    {"RANDRB", 4, 0, {0, 0}, 1, 0, 8, -1},
    {"RANDTB", 4, 0, {0, 0}, 0, 0, 4, -1},
    {"SEL2R", 4, 2, {0x95, 0x70}, 1, 1, 5, 0},
    {"SEL2T", 4, 1, {0x00, 0}, 0, 1, 0, -1},
    {"READR", 5, 1, {0x30, 0}, 1, 1, 1, -1},
    {"READT", 5, 0, {0, 0}, 0, 1, 16, -1},
    {"HALT", 10, 2, {0x50, 0x00}, 1, 1, 0, -1},
    {"WRITE", 6, 1, {0xA2, 0}, 1, 1, 5, -1},
    {"COMPW1", 6, 1, {0xA0, 0}, 1, 1, 1, -1},
    {"COMPW2", 7, 0, {0, 0}, 1, 1, 16, -1},
};

inline bool compatible(const CommandDef &c, const uint8_t *b, int n) {   // command.py:44-67
    if (c.total() != n) return false;
    for (int i = 0; i < c.n_header; i++)
        if (c.header[i] != b[i]) return false;
    int end = n;
    if (c.crc) {
        end -= 2;
        if (!crc_a_ok(b, (size_t)n)) return false;
    }
    if (c.xor_check >= 0) {
        int a = c.xor_check;
        for (int i = c.n_header; i < end; i++) a ^= b[i];
        if (a != 0) return false;
    }
    return true;
}

// candidates per protocol stage and direction, in the reference's order (command.py:120-137)
static const int TAG_STAGE[6][4] = {{CMD_ATQAUL, CMD_ATQA1K, CMD_ATQA4K, CMD_ATQADS}, {CMD_ANTI1U, CMD_ANTI1G, -1, -1},
                                    {CMD_SEL1U, CMD_SEL1K, -1, -1},                  {CMD_ANTI2T, CMD_RANDTA, -1, -1},
                                    {CMD_SEL2T, CMD_RANDTB, -1, -1},                 {CMD_READT, -1, -1, -1}};
// reader table: stages 0..7 and 10; the reference tests `stage < number of entries (9)` and then indexes by stage
static const int READER_STAGE[8][3] = {{CMD_REQA, CMD_WUPA, -1},          {CMD_ANTI1R, -1, -1},        {CMD_SEL1R, -1, -1},
                                       {CMD_ANTI2R, CMD_AUTHA, CMD_AUTHB}, {CMD_SEL2R, CMD_RANDRB, -1}, {CMD_READR, -1, -1},
                                       {CMD_WRITE, CMD_COMPW1, -1},        {CMD_COMPW2, -1, -1}};

inline int find_command(const uint8_t *b, int n, int type, int prev_cmd) {   // command.py:166-199
    const int ind = COMMANDS[prev_cmd].stage;
    for (int v = ind; v <= ind + 1; v++) {
        if (type == 0) {
            if (v < 6)
                for (int k = 0; k < 4 && TAG_STAGE[v][k] >= 0; k++)
                    if (compatible(COMMANDS[TAG_STAGE[v][k]], b, n)) return TAG_STAGE[v][k];
        } else if (v < 8) {   // (stage 8 would raise KeyError in the reference; no command has stage 7 + 1 as its successor in a trace)
            for (int k = 0; k < 3 && READER_STAGE[v][k] >= 0; k++)
                if (compatible(COMMANDS[READER_STAGE[v][k]], b, n)) return READER_STAGE[v][k];
        }
    }
    if (n < 1) return -1;
    // by leading bytes (command.py:139-164): the first listed option, or the second when the second byte names it
    int opt[2] = {-1, -1};
    switch (b[0]) {
    case 0x00: opt[0] = CMD_SEL2T; break;
    case 0x02: opt[0] = CMD_ATQA4K; break;
    case 0x03: opt[0] = CMD_ATQADS; break;
    case 0x04: opt[0] = CMD_SEL1U; opt[1] = CMD_ATQA1K; break;
    case 0x08: opt[0] = CMD_SEL1K; break;
    case 0x26: opt[0] = CMD_REQA; break;
    case 0x30: opt[0] = CMD_READR; break;
    case 0x44: opt[0] = CMD_ATQAUL; break;
    case 0x50: opt[0] = CMD_HALT; break;
    case 0x52: opt[0] = CMD_WUPA; break;
    case 0x60: opt[0] = CMD_AUTHA; break;
    case 0x61: opt[0] = CMD_AUTHB; break;
    case 0x88: opt[0] = CMD_ANTI1U; break;
    case 0x93: opt[0] = CMD_ANTI1R; opt[1] = CMD_SEL1R; break;
    case 0x95: opt[0] = CMD_ANTI2R; opt[1] = CMD_SEL2R; break;
    case 0xA0: opt[0] = CMD_COMPW1; break;
    case 0xA2: opt[0] = CMD_WRITE; break;
    default: return -1;
    }
    int option = opt[0];
    if (opt[1] >= 0) {
        if (n < 2) return -1;   // (IndexError in the reference: caught, no command)
        const int s = opt[1];
        bool named;
        switch (b[1]) {
        case 0x00: named = (s == CMD_ATQAUL || s == CMD_HALT || s == CMD_ATQA1K); break;
        case 0x20: named = (s == CMD_ANTI1R || s == CMD_ANTI2R); break;
        case 0x70: named = (s == CMD_SEL1R || s == CMD_SEL2R); break;
        default: return -1;     // (KeyError in the reference: caught, no command)
        }
        if (named) option = s;
    }
    return compatible(COMMANDS[option], b, n) ? option : -1;
}

// ---- CRYPTO1 (cipher.py) ----
// 48-bit shift register; bit i of `st` is the i-th oldest bit (cipher.py keeps the whole bit history and looks
// at its last 48).  One filter bit per clock; a clock shifts in L(st) ^ (input & feed_in) ^ (filter & feed_ks).
struct Crypto1 {
    uint64_t st = 0;
    uint8_t ar[4] = {0, 0, 0, 0}, at[4] = {0, 0, 0, 0};   // the expected reader / tag answers (suc64, suc96 of the tag nonce)

    void load_key(const uint8_t key[6]) {   // cipher.py:11-12: key bytes, least significant bit first
        st = 0;
        for (int i = 0; i < 48; i++) st |= (uint64_t)((key[i >> 3] >> (i & 7)) & 1) << i;
    }
    static int fa(int a, int b, int c, int d) { return ((a | b) ^ (a & d)) ^ (c & ((a ^ b) | d)); }   // cipher.py:97-99
    static int fb(int a, int b, int c, int d) { return ((a & b) | c) ^ ((a ^ b) & (c | d)); }          // cipher.py:101-103
    static int fc(int a, int b, int c, int d, int e) {                                                 // cipher.py:105-107
        return (a | ((b | e) & (d ^ e))) ^ ((a ^ (b & d)) & ((c ^ d) | (b & e)));
    }
    int bit(int i) const { return (int)((st >> i) & 1); }
    int filter() const {   // cipher.py:110-118
        const int a = fa(bit(9), bit(11), bit(13), bit(15)), b = fb(bit(17), bit(19), bit(21), bit(23));
        const int c = fb(bit(25), bit(27), bit(29), bit(31)), d = fa(bit(33), bit(35), bit(37), bit(39));
        const int e = fb(bit(41), bit(43), bit(45), bit(47));
        return fc(a, b, c, d, e);
    }
    int feedback() const {   // cipher.py:75-76
        static const int taps[18] = {0, 5, 9, 10, 12, 14, 15, 17, 19, 24, 25, 27, 29, 35, 39, 41, 42, 43};
        int l = 0;
        for (int t : taps) l ^= bit(t);
        return l;
    }
    // cipher.py:20-35: every bit is xored with the filter output; the ninth bit of a group (the parity, when
    // has_parity) reuses the keystream bit of the next data bit -- the register does not move for it
    void crypt(const uint8_t *in, size_t n, uint8_t *out, int feed_in, int feed_ks, int has_parity) {
        int i = 0;
        for (size_t k = 0; k < n; k++) {
            const int f = filter();
            const int b = in[k] & 1;
            out[k] = (uint8_t)(f ^ b);
            if (i < 8 || !has_parity) {
                const uint64_t nx = (uint64_t)(feedback() ^ (b & feed_in) ^ (f & feed_ks));
                st = (st >> 1) | (nx << 47);
                i++;
            } else {
                i = 0;
            }
        }
    }
    // cipher.py:37-42 with lfsr.py: the 32-bit nonce register, taps 16 18 19 21; ar after 64 clocks, at after 96
    void set_answers(const uint8_t nonce_bits[32]) {
        uint8_t r[32];
        memcpy(r, nonce_bits, 32);
        int idx = 0;
        auto advance = [&](int ticks) {
            for (int t = 0; t < ticks; t++) {
                const int b = r[(16 + idx) & 31] ^ r[(18 + idx) & 31] ^ r[(19 + idx) & 31] ^ r[(21 + idx) & 31];
                r[idx] = (uint8_t)b;
                idx = (idx + 1) & 31;
            }
        };
        auto bytes = [&](uint8_t out[4]) {
            for (int k = 0; k < 4; k++) {
                int v = 0;
                for (int i = 0; i < 8; i++) v |= (r[(idx + 8 * k + i) & 31] & 1) << i;
                out[k] = (uint8_t)v;
            }
        };
        advance(64);
        bytes(ar);
        advance(32);
        bytes(at);
    }
    // cipher.py:62-72: mix uid ^ tag nonce into the register.  Plain nonce (first authentication): 32 bits in,
    // nothing returned.  Encrypted nonce (nested authentication): 36 bits with parity in, the decrypted 36 out.
    bool set_tag(const std::vector<uint8_t> &uid, const uint8_t *nonce, size_t n_nonce, bool encrypted, uint8_t *plain_out) {
        uint8_t ub[64], xb[64], kb[64];
        size_t ll = 0;
        for (size_t k = 0; k < uid.size() && ll + 9 <= sizeof ub; k++) {
            for (int i = 0; i < 8; i++) ub[ll++] = (uint8_t)((uid[k] >> i) & 1);
            if (encrypted) ub[ll++] = 0;   // cipher.py:54-60: a zero where the parity bit sits
        }
        if (n_nonce < ll || ll == 0) return false;   // (IndexError in the reference)
        for (size_t i = 0; i < ll; i++) xb[i] = ub[i] ^ (nonce[i] & 1);
        crypt(xb, ll, kb, 1, encrypted ? 1 : 0, encrypted ? 1 : 0);
        uint8_t nb[32];
        size_t m = 0;
        for (size_t i = 0; i < ll; i++) {
            const uint8_t pb = encrypted ? (uint8_t)(ub[i] ^ kb[i]) : (uint8_t)(nonce[i] & 1);
            if (plain_out) plain_out[i] = pb;
            if ((!encrypted || i % 9 != 8) && m < 32) nb[m++] = pb;
        }
        if (m != 32) return false;
        set_answers(nb);
        return true;
    }
};

}  // namespace nfc

// ---- the machine (fsm.py) ----
struct nfc_fsm {
    nfc_fsm() = default;
    nfc_fsm(const nfc_fsm &o) { *this = o; }
    nfc_fsm &operator=(const nfc_fsm &o) {
        cur_cmd = o.cur_cmd; tag_type = o.tag_type; encrypted = o.encrypted; cipher = o.cipher; uid = o.uid;
        memcpy(key_a, o.key_a, 6); memcpy(key_b, o.key_b, 6);
        cur_key = (o.cur_key == o.key_b) ? key_b : key_a;
        return *this;
    }
    int cur_cmd = nfc::CMD_REQA;
    int tag_type = -1;          // -1 none, 0 Ultralight, 1 Classic 1K, 2 Classic 4K, 3 DESFire (command.py:70-74)
    int encrypted = 0;          // a CRYPTO1 session is up (fsm._encryption)
    nfc::Crypto1 cipher;
    uint8_t key_a[6] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF}, key_b[6] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};   // fsm.py:157-160
    const uint8_t *cur_key = key_a;
    std::vector<uint8_t> uid;
    void reset_tag() {   // fsm.py:20-24
        uid.clear();
        tag_type = -1;
        encrypted = 0;
    }
};

namespace nfc {

// One packet (bits as PacketProcessor hands them over, packets.py:94-98).  Returns the frame record; bytes_out
// receives the frame's bytes (capacity >= n_bits / 9 + 1).
// enc_out (optional, same capacity): while a session is up, what was on the air -- one entry per nine bits, the byte in
// the low half, bit 8 set when the parity bit equals the data parity (the '!' of fsm._print_enc, fsm.py:113-131).
inline void fsm_process(nfc_fsm &F, const uint8_t *bits_in, size_t n_bits, int type, nfc_frame *out, uint8_t *bytes_out,
                        uint16_t *enc_out = nullptr) {
    memset(out, 0, sizeof *out);
    out->type = type;
    out->cmd = NFC_CMD_UNKNOWN;
    // fsm.py:49-66: bring the bit count to a multiple of nine
    const int start_bit = (type == 0) ? 1 : 0;   // packets.py:24-28
    std::vector<uint8_t> bits(bits_in, bits_in + n_bits);
    const size_t rem = n_bits % 9;
    if (rem == 8) bits.push_back((uint8_t)start_bit);
    else if (rem == 1) {
        if (bits.back() != start_bit) out->flags |= NFC_FRAME_EXTRA_ERROR;
        bits.pop_back();
    } else if (rem != 0) {
        out->flags |= NFC_FRAME_MANY_MORE_ERROR;
        bits.resize(n_bits - rem);
    }
    if (F.encrypted) {   // fsm.py:133-154
        out->flags |= NFC_FRAME_ENCRYPTED;
        int ne = 0;
        {
            int cur = 0, ones = 0, k = 0;
            for (uint8_t bit : bits) {
                if (k < 8) {
                    cur |= (bit & 1) << k;
                    ones += bit & 1;
                    k++;
                } else {
                    if (enc_out) enc_out[ne] = (uint16_t)(cur | (((ones & 1) == (bit & 1)) ? 0x100 : 0));
                    ne++;
                    cur = ones = k = 0;
                }
            }
        }
        out->n_enc = (uint16_t)ne;
        std::vector<uint8_t> plain(bits.size());
        const size_t ls = (bits.size() + 1) / 9;
        if (F.cur_cmd == CMD_RANDTA && ls == (size_t)COMMANDS[CMD_RANDRB].total()) {
            // {nr}{ar}: the reader nonce is fed back into the register while it is decrypted
            const size_t ll = bits.size() / 2;
            F.cipher.crypt(bits.data(), ll, plain.data(), 1, 1, 1);
            F.cipher.crypt(bits.data() + ll, bits.size() - ll, plain.data() + ll, 0, 0, 1);
        } else if (F.cur_cmd == CMD_AUTHA || F.cur_cmd == CMD_AUTHB) {
            // nested authentication: a fresh register keyed for the new sector swallows the encrypted tag nonce
            F.cipher = Crypto1();
            F.cipher.load_key(F.cur_key);
            plain.assign(bits.size(), 0);
            size_t ll = F.uid.size() * 9;
            if (!F.cipher.set_tag(F.uid, bits.data(), bits.size(), true, plain.data())) ll = 0;
            plain.resize(ll);
        } else {
            F.cipher.crypt(bits.data(), bits.size(), plain.data(), 0, 0, 1);
        }
        bits.swap(plain);
    }
    // fsm.py:28-47: eight data bits LSB first, then the odd-parity bit
    int nb = 0;
    {
        int cur = 0, ones = 0, k = 0;
        bool bad = false;
        for (uint8_t bit : bits) {
            if (k < 8) {
                cur |= (bit & 1) << k;
                ones += bit & 1;
                k++;
            } else {
                if ((ones & 1) == (bit & 1)) { bad = true; break; }
                bytes_out[nb++] = (uint8_t)cur;
                cur = ones = k = 0;
            }
        }
        if (!bad && k == 8) bytes_out[nb++] = (uint8_t)cur;
        if (bad) nb = 0;
    }
    if (nb == 0) {   // `if not bytes` (fsm.py:225): a parity error, or nothing left
        out->cmd = NFC_CMD_PARITY_ERROR;
        return;
    }
    out->n_bytes = (uint16_t)nb;
    const int cmd = find_command(bytes_out, nb, type, F.cur_cmd);
    if (cmd >= 0) F.cur_cmd = cmd;
    out->cmd = cmd >= 0 ? cmd : NFC_CMD_UNKNOWN;
    // command.py:245-253
    if (cmd >= 0) {
        const CommandDef &c = COMMANDS[cmd];
        out->n_header = (uint16_t)c.n_header;
        out->n_crc = (uint16_t)(c.crc ? 2 : 0);
        out->n_extra = (uint16_t)(nb - c.n_header - (c.crc ? 2 : 0));
    } else {
        out->n_extra = (uint16_t)nb;
    }
    const uint8_t *extra = bytes_out + out->n_header;
    // fsm.py:165-216
    auto same_tail = [&](const uint8_t *u, size_t n) {
        return F.uid.size() >= n && memcmp(F.uid.data() + F.uid.size() - n, u, n) == 0;
    };
    switch (cmd) {
    case CMD_REQA: case CMD_WUPA: case CMD_HALT: F.reset_tag(); break;
    case CMD_ATQAUL: F.tag_type = 0; break;
    case CMD_ATQA1K: F.tag_type = 1; break;
    case CMD_ATQA4K: F.tag_type = 2; break;
    case CMD_ATQADS: F.tag_type = 3; break;
    case CMD_ANTI1U: F.uid.insert(F.uid.end(), extra, extra + 3); break;
    case CMD_ANTI1G: F.uid.insert(F.uid.end(), extra, extra + 4); break;
    case CMD_SEL1R: {
        const int start = F.tag_type == 0 ? 1 : 0;
        const size_t n = (size_t)(4 - start);
        if (!(F.uid.size() == n && memcmp(F.uid.data(), extra + start, n) == 0)) {
            out->flags |= NFC_FRAME_UID_MISMATCH;
            F.uid.assign(extra + start, extra + 4);
        }
        break;
    }
    case CMD_ANTI2T: F.uid.insert(F.uid.end(), extra, extra + 4); break;
    case CMD_SEL2R:
        if (!same_tail(extra, 4)) {
            out->flags |= NFC_FRAME_UID_MISMATCH;
            F.uid.insert(F.uid.end(), extra, extra + 4);
        }
        break;
    case CMD_AUTHA: F.cur_key = F.key_a; break;
    case CMD_AUTHB: F.cur_key = F.key_b; break;
    case CMD_RANDTA:
        if (!F.encrypted) {   // first authentication: the tag nonce came in the clear (fsm.py:197-202)
            F.cipher = Crypto1();
            F.cipher.load_key(F.cur_key);
            uint8_t nb[32];
            for (int i = 0; i < 32; i++) nb[i] = (uint8_t)((extra[i >> 3] >> (i & 7)) & 1);
            if (out->n_extra >= 4 && F.cipher.set_tag(F.uid, nb, 32, false, nullptr)) F.encrypted = 1;
        }
        break;
    case CMD_RANDRB:   // fsm.py:203-208
        out->flags |= (F.encrypted && memcmp(extra + 4, F.cipher.ar, 4) == 0) ? NFC_FRAME_AR_OK : NFC_FRAME_AR_ERROR;
        break;
    case CMD_RANDTB:   // fsm.py:209-214
        out->flags |= (F.encrypted && memcmp(extra, F.cipher.at, 4) == 0) ? NFC_FRAME_AT_OK : NFC_FRAME_AT_ERROR;
        break;
    default: break;
    }
}


// fsm.process_outgoing (fsm.py:68-112): what an EMULATOR is about to send.  The machine follows the frame as if it had been heard --
// tag type from an ATQA, the command in flight -- and, while a MIFARE Classic session is up, encrypts it: the frame's bits (parity
// bits included, as the encoders take them) in, the bits to put on the air out.  Returns 1 when the tag is an Ultralight: the
// reference then runs the frame through process_bits "to update state" (fsm.py:71-72) -- the caller does, with its callback.
inline int fsm_process_outgoing(nfc_fsm &F, const uint8_t *bits, size_t n, int cmd, uint8_t *out) {
    if (F.tag_type == 0) return 1;
    for (size_t i = 0; i < n; i++) out[i] = bits[i] & 1;
    if (F.tag_type == 1) {   // CLASSIC1K (fsm.py:73-99)
        F.cur_cmd = cmd;
        if (F.encrypted && cmd != CMD_RANDTA) {
            if (cmd == CMD_RANDRB) {   // {nr}{ar}: the reader's own nonce feeds the register as it is encrypted
                const size_t ll = n / 2;
                F.cipher.crypt(bits, ll, out, 1, 0, 1);
                F.cipher.crypt(bits + ll, n - ll, out + ll, 0, 0, 1);
            } else {
                F.cipher.crypt(bits, n, out, 0, 0, 1);
            }
        } else if (cmd == CMD_RANDTA) {
            // the tag's nonce: a fresh register keyed for the sector takes uid ^ nonce; under a session that is already up
            // (nested authentication) the OLD register encrypts what goes on the air
            Crypto1 old = F.cipher;
            const bool was = F.encrypted != 0;
            uint8_t nb[32];
            size_t m = 0;
            for (size_t i = 0; i < n && m < 32; i++)
                if (i % 9 != 8) nb[m++] = bits[i] & 1;
            F.cipher = Crypto1();
            F.cipher.load_key(F.cur_key);
            if (m == 32 && F.cipher.set_tag(F.uid, nb, 32, false, nullptr)) F.encrypted = 1;
            if (was) old.crypt(bits, n, out, 0, 0, 1);
        }
        return 0;
    }
    switch (cmd) {   // no tag type yet: an ATQA sets it (fsm.py:101-108)
    case CMD_ATQAUL: F.tag_type = 0; break;
    case CMD_ATQA1K: F.tag_type = 1; break;
    case CMD_ATQA4K: F.tag_type = 2; break;
    case CMD_ATQADS: F.tag_type = 3; break;
    default: break;
    }
    return 0;
}

}  // namespace nfc
