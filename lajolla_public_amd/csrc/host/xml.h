// Minimal XML DOM reader for the Mitsuba-0.x scene dialect (the reference uses pugixml, parse_scene.cpp:2).
// Handles: prolog, comments, CDATA-free element trees, single/double-quoted attributes, the five
// predefined entities and numeric character references.  Text nodes are skipped (the dialect keeps all data
// in attributes).
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace lj {

struct XmlNode {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlNode>> children;

    bool has(const char *key) const {
        for (auto &a : attrs) if (a.first == key) return true;
        return false;
    }
    // pugixml's attribute("x").value() yields "" for a missing attribute.
    const std::string &attr(const char *key) const {
        static const std::string empty;
        for (auto &a : attrs) if (a.first == key) return a.second;
        return empty;
    }
    const XmlNode *child(const char *nm) const {
        for (auto &c : children) if (c->name == nm) return c.get();
        return nullptr;
    }
};

class XmlParser {
public:
    explicit XmlParser(const std::string &text) : s_(text) {}
    std::unique_ptr<XmlNode> parse_document() {
        std::unique_ptr<XmlNode> root;
        for (;;) {
            skip_misc();
            if (pos_ >= s_.size()) break;
            if (s_[pos_] != '<') fail("text outside the root element");
            auto n = parse_element();
            if (!root) root = std::move(n);
        }
        if (!root) fail("no root element");
        return root;
    }

private:
    const std::string &s_;
    size_t pos_ = 0;

    [[noreturn]] void fail(const std::string &why) const {
        size_t line = 1;
        for (size_t i = 0; i < pos_ && i < s_.size(); i++) if (s_[i] == '\n') line++;
        throw std::runtime_error("XML parse error at line " + std::to_string(line) + ": " + why);
    }
    bool starts(const char *lit) const { return s_.compare(pos_, std::char_traits<char>::length(lit), lit) == 0; }
    void skip_ws() { while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\t' || s_[pos_] == '\n' || s_[pos_] == '\r')) pos_++; }
    void skip_until(const char *lit) {
        size_t e = s_.find(lit, pos_);
        if (e == std::string::npos) fail(std::string("unterminated construct, expected ") + lit);
        pos_ = e + std::char_traits<char>::length(lit);
    }
    // whitespace, comments, processing instructions, doctype, and stray text between elements
    void skip_misc() {
        for (;;) {
            while (pos_ < s_.size() && s_[pos_] != '<') pos_++;
            if (pos_ >= s_.size()) return;
            if (starts("<!--")) { pos_ += 4; skip_until("-->"); }
            else if (starts("<?")) { pos_ += 2; skip_until("?>"); }
            else if (starts("<!")) { pos_ += 2; skip_until(">"); }
            else return;
        }
    }
    static bool name_char(char c) {
        return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-' || c == ':' || c == '.';
    }
    std::string parse_name() {
        size_t b = pos_;
        while (pos_ < s_.size() && name_char(s_[pos_])) pos_++;
        if (pos_ == b) fail("expected a name");
        return s_.substr(b, pos_ - b);
    }
    std::string decode(const std::string &raw) const {
        std::string out; out.reserve(raw.size());
        for (size_t i = 0; i < raw.size(); i++) {
            if (raw[i] != '&') { out.push_back(raw[i]); continue; }
            size_t e = raw.find(';', i);
            if (e == std::string::npos) { out.push_back('&'); continue; }
            std::string ent = raw.substr(i + 1, e - i - 1);
            if (ent == "amp") out.push_back('&');
            else if (ent == "lt") out.push_back('<');
            else if (ent == "gt") out.push_back('>');
            else if (ent == "quot") out.push_back('"');
            else if (ent == "apos") out.push_back('\'');
            else if (!ent.empty() && ent[0] == '#') {
                unsigned long cp = (ent.size() > 1 && (ent[1] == 'x' || ent[1] == 'X')) ? std::stoul(ent.substr(2), nullptr, 16) : std::stoul(ent.substr(1));
                if (cp < 0x80) out.push_back((char)cp);
                else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
            } else { out += raw.substr(i, e - i + 1); }
            i = e;
        }
        return out;
    }
    std::unique_ptr<XmlNode> parse_element() {
        pos_++; // '<'
        auto node = std::make_unique<XmlNode>();
        node->name = parse_name();
        for (;;) {
            skip_ws();
            if (pos_ >= s_.size()) fail("unterminated start tag");
            if (starts("/>")) { pos_ += 2; return node; }
            if (s_[pos_] == '>') { pos_++; break; }
            std::string key = parse_name();
            skip_ws();
            if (pos_ >= s_.size() || s_[pos_] != '=') fail("expected '=' after attribute name");
            pos_++; skip_ws();
            if (pos_ >= s_.size() || (s_[pos_] != '"' && s_[pos_] != '\'')) fail("expected a quoted attribute value");
            char q = s_[pos_++];
            size_t e = s_.find(q, pos_);
            if (e == std::string::npos) fail("unterminated attribute value");
            node->attrs.emplace_back(std::move(key), decode(s_.substr(pos_, e - pos_)));
            pos_ = e + 1;
        }
        for (;;) {
            skip_misc();
            if (pos_ >= s_.size()) fail("missing end tag for <" + node->name + ">");
            if (starts("</")) {
                pos_ += 2;
                std::string nm = parse_name();
                if (nm != node->name) fail("mismatched end tag </" + nm + "> for <" + node->name + ">");
                skip_ws();
                if (pos_ >= s_.size() || s_[pos_] != '>') fail("malformed end tag");
                pos_++;
                return node;
            }
            node->children.push_back(parse_element());
        }
    }
};

} // namespace lj
