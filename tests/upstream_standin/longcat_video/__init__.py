"""TEST INFRASTRUCTURE: a pure-PyTorch stand-in for the un-vendored `meituan-longcat/LongCat-Video` package, built on
`oracle/` — i.e. an "upstream" that behaves exactly as `spec/dit.md` §A ASSUMES upstream behaves.  It exists so that
`tests/first_contact_guards.py` can be executed end to end (every guard must PASS against it, and FAIL when the stand-in is
bent by `STANDIN_BREAK`) before a real checkout is ever visible.  Never imported by the product; nothing here is a claim
about the real upstream code."""
