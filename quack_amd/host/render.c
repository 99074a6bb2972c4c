/*
 * render.c — post-processing and SVG output of the drop-in CLI (host, C).
 *
 * Not GPU work: O(max_length x 97) once per file.  It has to reproduce the
 * reference's stdout byte for byte, so every arithmetic quirk of
 *     transform()  quack.c:230-293
 *     draw()       quack.c:295-856   (+ macros quack.c:15-50)
 *     svg writer   svg.c:12-104
 * is restated here with the same C types (int sums, float averages, %d of
 * 64-bit counters, ...).  The writer itself is organised differently: one
 * printf-style call per element with the attribute list in the format string,
 * and an explicit writer object instead of a global indent level.
 */
#include "quack_host.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ writer */

static void put_indent(qkh_svg *w) {
  for (int i = 0; i < w->depth; i++) fputs("  ", w->out);   /* svg.c:8,66-67 */
}

/* <elem attrs> (container: children follow, depth grows)      svg.c:59-89 */
static void open_tag(qkh_svg *w, const char *elem, const char *attr_fmt, ...) {
  va_list ap;
  put_indent(w);
  w->depth++;
  fprintf(w->out, "<%s", elem);
  va_start(ap, attr_fmt);
  vfprintf(w->out, attr_fmt, ap);
  va_end(ap);
  fputs(">\n", w->out);
}

/* <elem attrs/> */
static void leaf_tag(qkh_svg *w, const char *elem, const char *attr_fmt, ...) {
  va_list ap;
  put_indent(w);
  fprintf(w->out, "<%s", elem);
  va_start(ap, attr_fmt);
  vfprintf(w->out, attr_fmt, ap);
  va_end(ap);
  fputs("/>\n", w->out);
}

/* </elem>                                                     svg.c:92-104 */
static void close_tag(qkh_svg *w, const char *elem) {
  if (w->depth > 0) w->depth--;
  put_indent(w);
  fprintf(w->out, "</%s>\n", elem);
}

/* ---- the three text helpers of quack.c:15-50 ---- */

static void axis_label(qkh_svg *w, int x, int y, int rot, const char *label) {
  open_tag(w, "text",
           " x=\"%d\" fill=\"#AAA\" y=\"%d\" font-family=\"sans-serif\""
           " font-size=\"15px\" text-anchor=\"middle\" transform=\"rotate(%d)\"",
           x, y, rot);
  fprintf(w->out, "%s\n", label);
  close_tag(w, "text");
}

static void axis_number(qkh_svg *w, int x, int y, const char *anchor, int number) {
  open_tag(w, "text",
           " x=\"%d\" fill=\"#AAA\" y=\"%d\" font-family=\"sans-serif\""
           " font-size=\"10px\" text-anchor=\"%s\"",
           x, y, anchor);
  fprintf(w->out, "%d\n", number);
  close_tag(w, "text");
}

static void center_label_open(qkh_svg *w, int x, int y, const char *fill) {
  open_tag(w, "text",
           " x=\"%d\" y=\"%d\" fill=\"%s\" font-family=\"sans-serif\""
           " font-size=\"15px\" font-weight=\"bold\" text-anchor=\"middle\"",
           x, y, fill);
}

static void panel_title(qkh_svg *w, int y, const char *fill, const char *title) {
  open_tag(w, "text",
           " y=\"%d\" fill=\"%s\" x=\"%d\" font-family=\"sans-serif\" font-size=\"15px\"",
           y, fill, 5);
  fprintf(w->out, "%s\n", title);
  close_tag(w, "text");
}

/* --------------------------------------------------------------- transform */

void qkh_transform(qk_base_info *bases, uint64_t *max_length_io,
                   uint64_t *original_max_length, uint64_t number_of_sequences,
                   FILE *err) {
  uint64_t max_length = *max_length_io;
  int i, j;
  *original_max_length = max_length;              /* quack.c:232 */

  if (max_length > 3000) {                        /* quack.c:234-262 */
    const int bin_size = 100;
    int src, dst = 0;
    fprintf(err, "Binning...\n");
    for (src = 1; (uint64_t)src < max_length; src++) {
      if (src % bin_size == 0) {
        /* a new bin starts: its content, scores and length are cleared, its
         * kmer_count is NOT (quack.c:243-249) */
        dst++;
        memset(bases[dst].content, 0, sizeof bases[dst].content);
        memset(bases[dst].scores, 0, sizeof bases[dst].scores);
        bases[dst].length_count = 0;
      }
      for (i = 0; i < QK_N_BASES; i++) bases[dst].content[i] += bases[src].content[i];
      for (i = 0; i < QK_N_SCORES; i++) bases[dst].scores[i] += bases[src].scores[i];
      bases[dst].length_count += bases[src].length_count;
      bases[dst].kmer_count += bases[src].kmer_count;
    }
    max_length = (uint64_t)dst;                   /* last bin dropped, quack.c:261 */
  }

  for (i = 1; (uint64_t)i < max_length; i++)      /* cumulative, quack.c:264-266 */
    bases[i].kmer_count += bases[i - 1].kmer_count;

  for (i = 0; (uint64_t)i < max_length; i++) {    /* quack.c:269-291 */
    int score_sum = 0;
    for (j = 0; j < QK_N_SCORES; j++) score_sum = (int)(score_sum + bases[i].scores[j]);
    if (score_sum != 0)
      for (j = 0; j < QK_N_SCORES; j++)
        bases[i].scores[j] = 100 * bases[i].scores[j] / (uint64_t)score_sum;
    {
      float lc = 100 * (float)bases[i].length_count / (float)number_of_sequences;
      float kc = 100 * (float)bases[i].kmer_count / (float)number_of_sequences;
      bases[i].length_count = (uint64_t)ceil(lc);
      bases[i].kmer_count = (uint64_t)ceil(kc);
    }
  }
  *max_length_io = max_length;
}

/* -------------------------------------------------------------------- draw */

/* growable string for the polylines */
typedef struct {
  char *s;
  size_t len, cap;
} strbuf;

static void sb_add(strbuf *b, const char *t) {
  size_t n = strlen(t);
  if (b->len + n + 1 > b->cap) {
    size_t cap = b->cap ? b->cap * 2 : 256;
    while (cap < b->len + n + 1) cap *= 2;
    char *ns = realloc(b->s, cap);
    if (!ns) abort();
    b->s = ns;
    b->cap = cap;
  }
  memcpy(b->s + b->len, t, n + 1);
  b->len += n;
}

static void percent_rects(qkh_svg *w, const qk_base_info *bases, int max_length, int use_kmer) {
  for (int x = 0; x < max_length; x++) {
    uint64_t v = use_kmer ? bases[x].kmer_count : bases[x].length_count;
    if (v > 0)
      leaf_tag(w, "rect",
               " x=\"%d\" y=\"%d\" width=\"%d\" height=\"%d\" stroke=\"none\" fill=\"steelblue\"",
               x, 0, 1, (int)v);
  }
}

void qkh_draw(qkh_svg *w, const qk_base_info *bases, uint64_t max_length_u64,
              uint64_t number_of_sequences, int position, int adapters_used) {
  const int max_length = (int)max_length_u64;
  const int mirrored = position == 1;
  int i, j, x, y;
  int max_score = 40;                              /* quack.c:326 */
  uint64_t number_of_bases = 0;
  uint64_t total_counts[QK_N_SCORES];
  float *averages = calloc((size_t)(max_length > 0 ? max_length : 1), sizeof(float));
  char tmp[32];
  if (!averages) abort();
  memset(total_counts, 0, sizeof total_counts);

  /* encoding inference (quack.c:303-322): phred33 as soon as any position has
   * a count in score bins 0..30 */
  const char *encoding = "phred64";
  int offset = 31;
  for (i = 0; i < max_length && offset; i++)
    for (j = 0; j < 31; j++)
      if (bases[i].scores[j] > 0) {
        encoding = "phred33";
        offset = 0;
        break;
      }

  for (i = 0; i < max_length; i++) {               /* quack.c:327-341 */
    int sum = 0;
    for (j = offset; j < QK_N_SCORES; j++) {
      if (bases[i].scores[j] > 0 && (j - offset) > max_score) max_score = j - offset;
      total_counts[j - offset] += bases[i].scores[j];
      sum = (int)(sum + (uint64_t)(j - offset) * bases[i].scores[j]);
    }
    number_of_bases++;
    averages[i] = (float)(sum / 100.0);
  }

  /* ---- file stats line (quack.c:345-362) ---- */
  open_tag(w, "text",
           " x=\"%d\" y=\"%d\" text-anchor=\"middle\" font-family=\"sans-serif\""
           " font-size=\"15px\" fill=\"#555\"",
           mirrored ? 835 : 355, 20);
  open_tag(w, "tspan", "");
  fprintf(w->out, "%d", (int)number_of_sequences);
  close_tag(w, "tspan");
  open_tag(w, "tspan", " fill=\"#888\"");
  fputs("&#160;reads with endcoding&#160;", w->out);
  close_tag(w, "tspan");
  open_tag(w, "tspan", "");
  fprintf(w->out, "%s", encoding);
  close_tag(w, "tspan");
  close_tag(w, "text");

  /* ---- rug plot group, horizontal ticks (quack.c:365-384) ---- */
  open_tag(w, "g", " transform=\"translate(%d %d)\"", 5, 25);
  x = mirrored ? 1000 : 100;
  for (i = 10; i < 100; i += 10) {
    y = 105 + i * 250 / 100;
    leaf_tag(w, "line",
             " x1=\"%d\" x2=\"%d\" y1=\"%d\" y2=\"%d\" stroke=\"black\" stroke-width=\"%f\"",
             x, x + 100, y, y, (i % 20 == 10) ? 1.0 : 0.5);
  }

  /* ---- vertical section (quack.c:388-405) ---- */
  open_tag(w, "g", " transform=\"translate(%d 0)\"", mirrored ? 610 : 130);
  y = adapters_used == 0 ? 400 : 500;
  for (i = 10; i < 100; i += 10) {
    x = i * 450 / 100;
    leaf_tag(w, "line",
             " x1=\"%d\" x2=\"%d\" y1=\"%d\" y2=\"%d\" stroke=\"black\" stroke-width=\"%f\"",
             x, x, 10, y, (i % 20 == 10) ? 1.0 : 0.5);
  }

  /* ---- base content (quack.c:412-523) ---- */
  open_tag(w, "g", " transform=\"translate(%d,%d) scale(%d, %d)\"", 0, 100, 1, -1);
  open_tag(w, "svg",
           " width=\"%d\" height=\"%d\" preserveAspectRatio=\"none\" viewBox=\"0 0 %d %d\"",
           450, 100, max_length, (int)number_of_sequences);
  leaf_tag(w, "rect", " width=\"100%%\" height=\"100%%\" fill=\"#CCC\"");
  {
    static const char *colors[4] = {"#648964", "#89bc89", "#84accf", "#5d7992"};
    strbuf pts[4];
    memset(pts, 0, sizeof pts);
    /* cumulative raw counts, stacked A,T,C,G; the running sum is an int in
     * the reference (quack.c:448-463) */
    y = 0;
    for (i = 0; i < 4; i++) {
      y = (int)(y + bases[0].content[i]);
      snprintf(tmp, sizeof tmp, "0,%d ", y);
      sb_add(&pts[i], tmp);
    }
    for (x = 0; x < max_length; x++) {
      y = 0;
      for (i = 0; i < 4; i++) {
        y = (int)(y + bases[x].content[i]);
        snprintf(tmp, 20, "%d.5,%d ", x, y);
        sb_add(&pts[i], tmp);
      }
    }
    y = 0;
    for (i = 0; i < 4; i++) {
      y = (int)(y + bases[max_length - 1].content[i]);
      snprintf(tmp, 20, "%d,%d ", max_length, y);
      sb_add(&pts[i], tmp);
    }
    for (i = 3; i >= 0; i--)
      leaf_tag(w, "polyline", " points=\"0,0 %s %d,0\" fill=\"%s\" stroke=\"none\"",
               pts[i].s, max_length, colors[i]);
    for (i = 0; i < 4; i++) free(pts[i].s);
    close_tag(w, "svg");
    close_tag(w, "g");

    panel_title(w, 95, "#CCC", "Base Content Percentage");
    if (!mirrored) {
      static const char *labels[4] = {"%A", "%T", "%C", "%G"};
      for (i = 0; i < 4; i++) {
        center_label_open(w, 465, 20 * (4 - i), colors[i]);
        fprintf(w->out, "%s", labels[i]);
        close_tag(w, "text");
      }
    }
  }
  if (!mirrored) {
    axis_label(w, -50, -5, -90, "Percent");
    axis_number(w, -5, 100, "end", 0);
    axis_number(w, -5, 5, "end", 100);
  } else {
    axis_label(w, 50, -455, 90, "Percent");
    axis_number(w, 455, 100, "start", 0);
    axis_number(w, 455, 5, "start", 100);
  }

  /* ---- heat map + mean line (quack.c:528-628) ---- */
  open_tag(w, "g", " transform=\"translate(%d,%d) scale(%d, %d)\"", 0, 355, 1, -1);
  open_tag(w, "svg",
           " width=\"%d\" height=\"%d\" preserveAspectRatio=\"none\" viewBox=\"0 0 %d %d\"",
           450, 250, max_length, max_score);
  {
    static const char *band_fill[3] = {"#ccebc5", "#ffffcc", "#fbb4ae"};
    const int band_top[3] = {max_score, 28, 20};
    strbuf mean;
    memset(&mean, 0, sizeof mean);
    for (i = 0; i < 3; i++)
      leaf_tag(w, "rect",
               " x=\"%d\" y=\"%d\" width=\"100%%\" height=\"%d\" stroke=\"none\" fill=\"%s\"",
               0, 0, band_top[i], band_fill[i]);
    snprintf(tmp, sizeof tmp, "0,%0.2f ", averages[0]);
    sb_add(&mean, tmp);
    for (x = 0; x < max_length; x++) {
      for (y = 0; y < max_score; y++)
        if (bases[x].scores[y + offset] > 0)
          leaf_tag(w, "rect",
                   " x=\"%d\" y=\"%d\" fill-opacity=\"%f\" width=\"%d\" height=\"%d\""
                   " stroke=\"none\" stroke-width=\"%d\" fill=\"black\"",
                   x, y, (float)(bases[x].scores[y + offset]) / 100.0, 1, 1, 0);
      snprintf(tmp, 20, "%d.5,%0.2f ", x, averages[x]);
      sb_add(&mean, tmp);
    }
    snprintf(tmp, 20, "%d,%0.2f", max_length, averages[max_length - 1]);
    sb_add(&mean, tmp);
    leaf_tag(w, "polyline",
             " points=\"%s\" stroke=\"black\" stroke-width=\"%f\" stroke-opacity=\"%f\""
             " fill=\"none\" stroke-linejoin=\"round\"",
             mean.s, 0.5, 0.5);
    free(mean.s);
  }
  close_tag(w, "svg");
  close_tag(w, "g");
  panel_title(w, 350, "#888", "Per Base Sequence Quality");
  if (!mirrored) {
    const int mark[3] = {max_score, 28, 20};
    for (i = 0; i < 3; i++) {
      int yy = i == 0 ? 112 : 112 + (int)((max_score - mark[i]) * 250 / max_score);
      center_label_open(w, 465, yy, "#888");
      fprintf(w->out, "%d", mark[i]);
      close_tag(w, "text");
    }
  }

  /* ---- length distribution (quack.c:635-688) ---- */
  open_tag(w, "svg",
           " x=\"%d\" y=\"%d\" width=\"%d\" height=\"%d\" preserveAspectRatio=\"none\""
           " viewBox=\"0 0 %d 100\"",
           0, 360, 450, 100, max_length);
  leaf_tag(w, "rect", " width=\"100%%\" height=\"100%%\" fill=\"#EEE\"");
  percent_rects(w, bases, max_length, 0);
  close_tag(w, "svg");
  panel_title(w, 455, "#888", "Length Distribution");
  if (!mirrored) {
    axis_label(w, -410, -5, -90, "Percent");
    axis_number(w, -5, 370, "end", 0);
    axis_number(w, -5, 460, "end", 100);
  } else {
    axis_label(w, 410, -455, 90, "Percent");
    axis_number(w, 455, 370, "start", 0);
    axis_number(w, 455, 460, "start", 100);
  }

  /* ---- adapter distribution (quack.c:693-747) ---- */
  if (adapters_used == 1) {
    open_tag(w, "svg",
             " x=\"%d\" y=\"%d\" width=\"%d\" height=\"%d\" preserveAspectRatio=\"none\""
             " viewBox=\"0 0 %d 100\"",
             0, 465, 450, 100, max_length);
    leaf_tag(w, "rect", " width=\"100%%\" height=\"100%%\" fill=\"#EEE\"");
    percent_rects(w, bases, max_length, 1);
    close_tag(w, "svg");
    panel_title(w, 560, "#888", "Adapter Distribution");
    if (!mirrored) {
      axis_label(w, -515, -5, -90, "Percent");
      axis_number(w, -5, 475, "end", 0);
      axis_number(w, -5, 565, "end", 100);
    } else {
      axis_label(w, 515, -455, 90, "Percent");
      axis_number(w, 455, 475, "start", 0);
      axis_number(w, 455, 565, "start", 100);
    }
  }

  /* ---- bottom axis (quack.c:750-758) ---- */
  y = 470 + (adapters_used == 1 ? 105 : 0);
  axis_label(w, 225, y + 5, 0, "Base Pairs");
  axis_number(w, 0, y, "middle", 0);
  axis_number(w, 450, y, "middle", max_length);
  close_tag(w, "g");

  /* ---- score distribution (quack.c:766-852) ---- */
  open_tag(w, "g", " transform=\"translate(%d,%d) scale(%d, %d)\"",
           mirrored ? 1065 : 125, 355, mirrored ? 1 : -1, -1);
  open_tag(w, "svg",
           " width=\"%d\" height=\"%d\" preserveAspectRatio=\"none\" viewBox=\"0 0 100 %d\"",
           100, 250, max_score);
  leaf_tag(w, "rect", " width=\"100%%\" height=\"100%%\" fill=\"#EEE\"");
  for (y = 0; y < max_score; y++)
    if (total_counts[y] > 0)
      leaf_tag(w, "rect",
               " x=\"%d\" y=\"%d\" width=\"%d\" height=\"%d\" stroke=\"none\" fill=\"steelblue\"",
               0, y, (int)(total_counts[y] / number_of_bases), 1);
  close_tag(w, "svg");
  close_tag(w, "g");
  {
    const int tx = mirrored ? 1070 : 30;
    open_tag(w, "text",
             " y=\"%d\" fill=\"#888\" x=\"%d\" font-family=\"sans-serif\" font-size=\"15px\"",
             335, tx);
    open_tag(w, "tspan", "");
    fputs("Score\n", w->out);
    close_tag(w, "tspan");
    open_tag(w, "tspan", " dy=\"%d\" x=\"%d\"", 15, tx);
    fputs("Distribution\n", w->out);
    close_tag(w, "tspan");
    close_tag(w, "text");
  }
  if (!mirrored) {
    axis_label(w, 72, 100, 0, "Percent");
    axis_number(w, 25, 100, "middle", 100);
    axis_label(w, -230, 20, -90, "Score");
    axis_number(w, 20, 110, "end", max_score);
    axis_number(w, 20, 355, "end", 1);
  } else {
    axis_label(w, 1115, 100, 0, "Percent");
    axis_number(w, 1165, 100, "middle", 100);
    axis_label(w, 230, -1170, 90, "Score");
    axis_number(w, 1170, 110, "start", max_score);
    axis_number(w, 1170, 355, "start", 1);
  }
  close_tag(w, "g");
  free(averages);
}

/* ---------------------------------------------------- document envelope */

void qkh_svg_begin(qkh_svg *w, FILE *out, int paired, int adapters, const char *name) {
  const int width = paired ? 1195 : 615;            /* quack.c:879-883 */
  int height = adapters ? 610 : 510;
  if (name) height += 30;
  w->out = out;
  w->depth = 0;
  w->has_name = name != NULL;
  open_tag(w, "svg",
           " width=\"%d\" height=\"%d\" viewBox=\"%d %d %d %d\""
           " xmlns=\"http://www.w3.org/2000/svg\" xmlns:xlink=\"http://www.w3.org/1999/xlink\"",
           width, height, 0, 0, width, height);
  if (name) {                                       /* quack.c:894-908 */
    open_tag(w, "text",
             " x=\"%d\" y=\"%d\" font-family=\"sans-serif\" text-anchor=\"middle\""
             " font-size=\"30px\" fill=\"black\"",
             width / 2, 30);
    fprintf(w->out, "%s", name);
    close_tag(w, "text");
    open_tag(w, "g", " transform=\"translate(%d %d)\"", 0, 30);
  }
}

void qkh_svg_end(qkh_svg *w) {
  if (w->has_name) close_tag(w, "g");               /* quack.c:923-925 */
  close_tag(w, "svg");
}
