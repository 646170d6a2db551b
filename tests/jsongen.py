"""Document and number generators for the stage-2 (tape) tests: random valid JSON with every token kind, token-level
mutations that break it in the ways the reference's walk distinguishes, and number texts that sit on the rounding and range
boundaries of binary64 (all generated here, nothing read from the reference)."""
import numpy as np


def number_corner_cases():
    """texts of JSON numbers (valid and invalid), the hard ones for a decimal -> binary64 conversion among them"""
    out = ["0", "-0", "1", "-1", "10", "123456789", "9223372036854775807", "9223372036854775808", "-9223372036854775808", "-9223372036854775809",
           "18446744073709551615", "18446744073709551616", "99999999999999999999", "10000000000000000000", "12345678901234567890",
           "123456789012345678901234567890", "-123456789012345678901", "0.0", "-0.0", "0.5", "1.5", "0.1", "0.2", "0.3", "1e0", "1E0", "1e+0", "1e-0",
           "1e1", "1e22", "1e23", "1e-22", "1e-23", "9007199254740991", "9007199254740992", "9007199254740993", "9007199254740993.0", "9007199254740992.5",
           "9007199254740992.500000000000000000000000000000000001", "9007199254740994.5", "9007199254740993e0", "1.7976931348623157e308",
           "1.7976931348623158e308", "1.7976931348623159e308", "1.797693134862315807e308", "1.797693134862315808e308", "1.8e308", "1e308", "1e309", "2e308", "-1e309",
           "4.9406564584124654e-324", "4.9e-324", "5e-324", "2.5e-324", "2.4e-324", "2.4703282292062327e-324", "2.4703282292062328e-324", "2.4703282292062329e-324",
           "2.2250738585072014e-308", "2.2250738585072011e-308", "2.2250738585072012e-308", "2.2250738585072013e-308", "2.225073858507201e-308",
           "1e-323", "1e-324", "1e-325", "1e-342", "1e-343", "1e-400", "0e999999999999999999999999", "0.0e-999", "-0.0e-999", "1e999999999999999999999", "1e-999999999999999999999",
           "0e0", "0.0000000000000000000000000000000000001", "0." + "0" * 400 + "1", "1" + "0" * 400, "1" + "0" * 308, "1" + "0" * 309, "0." + "0" * 323 + "49", "0." + "0" * 323 + "25",
           "7.2057594037927933e16", "3.1415926535897932384626433832795028841971693993751", "10000000000000000000000000000000000000000000e+308", "1e-10000", "1e+10000",
           "123.", "1.e5", ".5", "-", "-.5", "01", "-01", "00", "0x10", "1e", "1e+", "1e-", "1ee5", "1.5.5", "1a", "12x", "1.5x", "1e5x", "--1", "+1", "1-", "-a", "1e5.5",
           "0.1e", "1E+", "2e+-3", "12345678901234567890x", "123456789012345678901234x", "-12345678901234567890", "1e000000000000000000001", "1e-000000000000000000001",
           "1.0000000000000000000000000000000000000000000000001", "0.99999999999999999999999999999999999999", "8.98846567431158e307", "8.988465674311581e307",
           "17976931348623157" + "0" * 292, "17976931348623158" + "0" * 292, "179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365510704342711559699508093042880177904174497791", "179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365510704342711559699508093042880177904174497792"]
    # exact halfway points between neighbouring doubles, and their neighbours one unit in the last written place
    from fractions import Fraction
    rng = np.random.default_rng(1234)
    for _ in range(60):
        m = int(rng.integers(1 << 52, 1 << 53))
        e = int(rng.integers(-1100, 960))
        half = Fraction(2 * m + 1, 1) * (Fraction(2) ** (e - 1)) if e - 1 >= 0 else Fraction(2 * m + 1, 2 ** (1 - e))
        # exact decimal expansion of a dyadic rational
        num, den = half.numerator, half.denominator  # den = 2^k
        k = den.bit_length() - 1
        digits = str(num * (5 ** k))  # half = digits * 10^-k
        for delta in (0, 1, -1):
            d = str(int(digits) + delta)
            if k == 0:
                text = d
            elif len(d) > k:
                text = d[:-k] + "." + d[-k:]
            else:
                text = "0." + "0" * (k - len(d)) + d
            out.append(text)
            if len(d) > 25:
                out.append(d[0] + "." + d[1:] + "e" + str(len(d) - 1 - k))
    # subnormal halves: (2m + 1) * 2^-1075
    for m in (0, 1, 2, 3, (1 << 52) - 1, (1 << 52) - 2, 12345678901234):
        num = (2 * m + 1) * (5 ** 1075)
        d = str(num)
        for delta in (0, 1, -1):
            dd = str(int(d) + delta)
            out.append("0." + "0" * (1075 - len(dd)) + dd)
    # random decimals of every length and exponent
    for _ in range(600):
        nd = int(rng.integers(1, 45))
        digits = "".join(str(int(x)) for x in rng.integers(0, 10, nd))
        digits = digits.lstrip("0") or "0"
        dot = int(rng.integers(0, len(digits) + 1))
        text = digits if dot == len(digits) or rng.random() < 0.3 else (digits[:dot] or "0") + "." + digits[dot:]
        if text.startswith("0") and len(text) > 1 and text[1] != ".":
            text = text.lstrip("0") or "0"
            if text.startswith("."):
                text = "0" + text
        if rng.random() < 0.6:
            text += ["e", "E"][int(rng.integers(0, 2))] + ["", "+", "-"][int(rng.integers(0, 3))] + str(int(rng.integers(0, 400)))
        if rng.random() < 0.3:
            text = "-" + text
        out.append(text)
    return out


def random_value(rng, depth=0, max_depth=6):
    def rstr():
        parts = []
        for _ in range(int(rng.integers(0, 6))):
            k = int(rng.integers(0, 12))
            parts.append(["a", "Zq", " ", "\\n", "\\\"", "\\\\", "\\/", "\\u00e9", "\\ud83d\\ude00", "日本", "\\t", "x" * int(rng.integers(1, 40))][k])
        return '"' + "".join(parts) + '"'

    def rnum():
        k = int(rng.integers(0, 9))
        if k == 0:
            return str(int(rng.integers(-1000, 1000)))
        if k == 1:
            return str(int(rng.integers(0, 1 << 62)) * (1 if rng.random() < 0.5 else -1))
        if k == 2:
            return repr(float(rng.random()))
        if k == 3:
            return f"{rng.random():.6g}"
        if k == 4:
            return f"{rng.random() * 10 ** int(rng.integers(-30, 30)):.17g}"
        if k == 5:
            return str(int(rng.integers(1 << 63, (1 << 64) - 1, dtype=np.uint64)))
        if k == 6:
            return f"{int(rng.integers(0, 1 << 40))}.{int(rng.integers(0, 1 << 40))}e{int(rng.integers(-320, 290))}"
        if k == 7:
            return "".join(str(int(x)) for x in rng.integers(1, 10, int(rng.integers(18, 30)))) + "." + "".join(str(int(x)) for x in rng.integers(0, 10, int(rng.integers(1, 30))))
        return ["0", "-0", "0.0", "1e5", "-1E-5", "1.5e+10"][int(rng.integers(0, 6))]

    ws = ["", "", "", " ", "\n", "  ", "\t", "\r\n"]
    w = lambda: ws[int(rng.integers(0, len(ws)))]
    kind = int(rng.integers(0, 2)) if depth == 0 and rng.random() < 0.9 else (int(rng.integers(0, 8)) if depth < max_depth else int(rng.integers(2, 8)))
    if kind == 0:
        return "[" + w() + ("," + w()).join(random_value(rng, depth + 1, max_depth) + w() for _ in range(int(rng.integers(0, 6)))) + "]"
    if kind == 1:
        return "{" + w() + ("," + w()).join(rstr() + w() + ":" + w() + random_value(rng, depth + 1, max_depth) + w() for _ in range(int(rng.integers(0, 6)))) + "}"
    if kind == 2:
        return rstr()
    if kind in (3, 4):
        return rnum()
    return ["true", "false", "null"][kind - 5]


def random_document(rng, max_depth=6):
    ws = ["", "", " ", "\n"]
    return (ws[int(rng.integers(0, 4))] + random_value(rng, 0, max_depth) + ws[int(rng.integers(0, 4))]).encode()


_TOKENS = [b",", b":", b"[", b"]", b"{", b"}", b'"k"', b"1", b"true", b"null", b"x", b"-", b'"\\q"', b"tru", b"nul", b"fals", b"1.", b"01", b" ", b"1e999", b"123456789012345678901"]


def mutate(rng, doc):
    """break (or not) a valid document at the token level: delete, insert, replace or duplicate a piece at a random place"""
    a = bytearray(doc)
    for _ in range(int(rng.integers(1, 4))):
        if not a:
            break
        pos = int(rng.integers(0, len(a)))
        k = int(rng.integers(0, 5))
        tok = _TOKENS[int(rng.integers(0, len(_TOKENS)))]
        if k == 0:
            del a[pos:pos + int(rng.integers(1, 4))]
        elif k == 1:
            a[pos:pos] = tok
        elif k == 2:
            a[pos:pos + 1] = tok
        elif k == 3:
            a = a[:pos]  # truncate
        else:
            a[pos:pos] = a[pos:pos + int(rng.integers(1, 6))]
    return bytes(a)
