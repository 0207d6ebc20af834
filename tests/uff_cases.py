"""Header variants for the .uff probe tests: name -> header text (bytes).  Shared by tests/test_iqfile_probe.py and
tests/golden/make_uff_golden.py."""


def cases():
    def doc(sample, blocks, root="SDR", pre='<?xml version="1.0" encoding="UTF-8"?>\n'):
        return (pre + "<%s>\n<Recorder Name='x' Version=\"1\"/>\n<Sample>%s</Sample>\n<Datablocks>%s</Datablocks>\n</%s>\n"
                % (root, sample, blocks, root)).encode()

    ch = lambda bits, cont, order, chans="<Channel Value=\"I\"/><Channel Value=\"Q\"/>", amount="2": (  # noqa: E731
        '<Channels Bits="%s" Container="%s" Ordering="%s" Amount="%s">%s</Channels>' % (bits, cont, order, amount, chans))
    sr = lambda val, unit: '<Samplerate Unit="%s" Value="%s"/>' % (unit, val)   # noqa: E731
    blk = lambda n: '<Datablock Count="%d" Number="1" Channel="Channel"><Frequency Value="1" Unit="KHz"/></Datablock>' % n   # noqa: E731
    c = {}
    for cont, bits in (("int8", 8), ("uint8", 8), ("int16", 16), ("int16", 12), ("int24", 24), ("int32", 32), ("float32", 32)):
        for order in ("MSB", "LSB"):
            c["std_%s_%d_%s" % (cont, bits, order)] = doc(sr(2048000, "Hz") + ch(bits, cont, order), blk(4000))
    c["unit_khz"] = doc(sr(2000, "KHz") + ch(16, "int16", "LSB"), blk(100))
    c["unit_khz_lower_h"] = doc(sr(2500, "Khz") + ch(16, "int16", "LSB"), blk(100))
    c["unit_mhz"] = doc(sr(2, "MHz") + ch(16, "int16", "LSB"), blk(100))
    c["unit_unknown_means_mega"] = doc(sr(2, "kHz") + ch(16, "int16", "LSB"), blk(100))
    c["rate_defaults"] = doc("<Samplerate/>" + ch(16, "int16", "LSB"), blk(100))
    c["channels_defaults"] = doc(sr(2048000, "Hz") + "<Channels><Channel Value='I'/><Channel Value='Q'/></Channels>", blk(100))
    c["no_sample_element"] = ('<?xml version="1.0"?><SDR><Datablocks>%s</Datablocks></SDR>' % blk(10)).encode()
    c["order_qi"] = doc(sr(2048000, "Hz") + ch(16, "int16", "LSB", '<Channel Value="Q"/><Channel Value="I"/>'), blk(100))
    c["order_i_only"] = doc(sr(2048000, "Hz") + ch(16, "int16", "LSB", '<Channel Value="I"/>', "1"), blk(100))
    c["order_q_only"] = doc(sr(2048000, "Hz") + ch(16, "int16", "LSB", '<Channel Value="Q"/>', "1"), blk(100))
    c["order_ii"] = doc(sr(2048000, "Hz") + ch(16, "int16", "LSB", '<Channel Value="I"/><Channel Value="I"/>'), blk(100))
    c["three_channels_listed"] = doc(sr(2048000, "Hz") + ch(16, "int16", "LSB", '<Channel Value="I"/><Channel Value="Q"/><Channel Value="I"/>'), blk(100))
    c["channel_value_default"] = doc(sr(2048000, "Hz") + ch(16, "int16", "LSB", '<Channel/><Channel Value="Q"/>'), blk(100))
    c["comments_everywhere"] = doc("<!-- a --> " + sr(2048000, "Hz") + "<!-- b -->" + ch(8, "uint8", "N/A") + "<!--c-->", "<!-- d -->" + blk(7) + "<!-- e -->" + blk(9))
    c["two_datablocks"] = doc(sr(2048000, "Hz") + ch(16, "int16", "MSB"), blk(1000) + blk(234))
    c["no_datablock"] = doc(sr(2048000, "Hz") + ch(16, "int16", "MSB"), "")
    c["datablock_count_default"] = doc(sr(2048000, "Hz") + ch(16, "int16", "MSB"), "<Datablock/>")
    c["other_root_name"] = doc(sr(2048000, "Hz") + ch(16, "int16", "MSB"), blk(50), root="Anything")
    c["no_xml_declaration"] = doc(sr(2048000, "Hz") + ch(16, "int16", "MSB"), blk(50), pre="")
    c["samplerate_outside_sample"] = ('<?xml version="1.0"?><SDR>%s<Sample>%s</Sample><Datablocks>%s</Datablocks></SDR>'
                                      % (sr(1234567, "Hz"), ch(16, "int16", "MSB"), blk(50))).encode()
    c["prefix_tag_names"] = doc('<SamplerateX Unit="Hz" Value="99"/>' + sr(2048000, "Hz") + ch(16, "int16", "MSB"), '<DatablockY Count="5"/>' + blk(50))
    c["unknown_container"] = doc(sr(2048000, "Hz") + ch(16, "int12", "MSB"), blk(50))
    c["malformed_unclosed"] = doc(sr(2048000, "Hz") + ch(16, "int16", "MSB"), blk(50)).replace(b"</Sample>", b"")
    c["malformed_mismatched_end"] = doc(sr(2048000, "Hz") + ch(16, "int16", "MSB"), blk(50)).replace(b"</Datablocks>", b"</Datablock>")
    c["malformed_unquoted_attr"] = doc('<Samplerate Unit=Hz Value="2048000"/>' + ch(16, "int16", "MSB"), blk(50))
    c["malformed_trailing_garbage"] = doc(sr(2048000, "Hz") + ch(16, "int16", "MSB"), blk(50)) + b"<oops"
    c["attribute_whitespace_and_newlines"] = doc('<Samplerate\n   Unit = "Hz"\n Value\t=\t\'2048000\' />' + ch(16, "int16", "MSB"), blk(50))
    c["entities_in_attribute"] = doc(sr(2048000, "Hz") + ch(16, "int16", "MSB").replace('Container="int16"', 'Container="int16" Note="a &amp; b &lt; c"'), blk(50))
    c["empty_file_header"] = b""
    c["not_xml_at_all"] = b"<?xml but then nothing sensible"
    return c
